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


def _lm_spike_loop(boom, X, y, prior, niter, seed, chains, lookahead=1, devices=()):
    """the body of lm_spike.__init__, boom.* calls only"""
    model = boom.RegressionModel(X, y, False, chains=chains, seed=seed, devices=list(devices))
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


def test_lm_spike_driver_loop_over_a_device_list(oracle):
    """devices=[0, 0]: `chains` chains on each entry; chain 0 and the last global chain are
    the oracle's chains with those ids (beta within 1e-8: the row-sharded X'X sums in
    another order)."""
    import boom_amd._boom as boom
    n, p, nsig, niter, seed, per = 500, 20, 4, 40, 4711, 4
    X, y, _ = regression_data(n, p, nsig, seed=22)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, nsig)
    model, sampler, draws, sd = _lm_spike_loop(boom, X, y, prior, niter, seed, per, 10, devices=(0, 0))
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    o = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", seed, 0), g0, niter)
    for i in range(niter):
        assert np.array_equal(draws[i] != 0, o["gamma"][i] != 0), i
        err = np.max(np.abs(draws[i] - o["beta"][i]) / np.maximum(np.abs(o["beta"][i]), 1e-3))
        assert err < 1e-8, (i, err)
    G, B, S = model.chain_states()
    assert G.shape == (2 * per, p) and S.shape == (2 * per,)
    ol = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", seed, 2 * per - 1), g0, niter)
    assert np.array_equal(G[-1], ol["gamma"][-1])
    assert abs(S[-1] - ol["sigsq"][-1]) < 1e-8 * S[-1]


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


@pytest.mark.parametrize("kind", ["logit", "probit"])
def test_binomial_spike_slab_on_the_pybind_module(kind):
    """boom.BinomialLogitModel(xdim, include_all).add_dataset(...) +
    boom.BinomialLogitSpikeSlabSampler(model, slab, spike, clt_threshold) (BayesBoom's
    call shapes, GlmModel_def.cpp:549-590, :966-1010): the loop's draws are those of the
    engine driven through the C-ABI with the same seed (which the oracle checks)."""
    import boom_amd
    import boom_amd._boom as boom
    from cases import logit_data, probit_data, probit_slab
    n, p, nsig, niter, seed, chains = 300, 9, 3, 10, 77, 4
    X, y, nt, _ = (logit_data if kind == "logit" else probit_data)(n, p, nsig, seed=4, max_trials=3)
    slab, pi = probit_slab(X, nt, nsig)
    Model = boom.BinomialLogitModel if kind == "logit" else boom.BinomialProbitModel
    Sampler = boom.BinomialLogitSpikeSlabSampler if kind == "logit" else boom.BinomialProbitSpikeSlabSampler
    model = Model(p, False, chains=chains, seed=seed)
    model.add_dataset(y, nt, X)
    sampler = Sampler(model, boom.MvnModel(slab["mu"], slab["prec"], True),
                      boom.VariableSelectionPrior(pi), 5)
    model.set_method(sampler)
    assert list(model.coef.inc) == [True] + [False] * (p - 1)
    draws = np.zeros((niter, p))
    for i in range(niter):
        model.sample_posterior()
        draws[i] = model.coef.Beta
    eng = boom_amd.Engine(chains, seed=seed)
    (eng.logit_set_data if kind == "logit" else eng.probit_set_data)(X, y, nt, 5)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    for i in range(niter):
        (eng.logit_sweep if kind == "logit" else eng.probit_sweep)(1)
        assert np.array_equal(draws[i], eng.get_state(0)[1]), i
    G, B = model.chain_states()
    assert np.array_equal(G, eng.get_states()[0]) and np.array_equal(B, eng.get_states()[1])


def test_structural_time_series_on_the_pybind_module():
    """boom.StateSpaceRegressionModel + LocalLinearTrendStateModel + SeasonalStateModel +
    StateSpacePosteriorSampler: same draws as the engine through the C-ABI."""
    import boom_amd
    import boom_amd._boom as boom
    from cases import bsts_priors, structural_data, structural_spec
    T, p, seed, chains, niter = 120, 4, 5, 3, 6
    X, y, _, obs = structural_data(T, p, 2, 4, seed=3, missing_frac=0.05)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, 2, 4)
    model = boom.StateSpaceRegressionModel(y, X, [bool(o) for o in obs], chains=chains, seed=seed)
    trend = boom.LocalLinearTrendStateModel()
    trend.set_initial_sigma(spec["var_initial_sigma"][0], spec["var_initial_sigma"][1])
    for i in range(2):
        trend.set_prior(i, spec["var_df"][i], spec["var_sigma_guess"][i], spec["var_sigma_upper_limit"][i])
    trend.set_initial_state_mean(spec["initial_state_mean"][:2])
    trend.set_initial_state_variance(spec["initial_state_variance"][:2])
    seas = boom.SeasonalStateModel(4)
    seas.set_sigsq(spec["var_initial_sigma"][2] ** 2)
    seas.set_prior(spec["var_df"][2], spec["var_sigma_guess"][2], spec["var_sigma_upper_limit"][2])
    seas.set_initial_state_mean(spec["initial_state_mean"][2:])
    seas.set_initial_state_variance(spec["initial_state_variance"][2])
    model.add_state(trend)
    model.add_state(seas)
    sampler = boom.StateSpacePosteriorSampler(model, boom.MvnGivenScalarSigma(prior["b"], prior["ominv"]),
                                              boom.ChisqModel(prior["df"], prior["sigma_guess"]),
                                              boom.VariableSelectionPrior(prior["pi"]), sig_up)
    model.set_method(sampler)
    assert model.state_dimension == 5
    for _ in range(niter):
        model.sample_posterior()
    eng = boom_amd.Engine(chains, seed=seed)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                   sigma_upper_limit=sig_up)
    eng.ss_set_structural(2, 4, spec["var_df"], spec["var_sigma_guess"], spec["var_sigma_upper_limit"],
                          spec["var_initial_sigma"], spec["initial_state_mean"], spec["initial_state_variance"])
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_sweep(niter)
    G, B, S = model.chain_states()
    g, b, s = eng.get_states()
    assert np.array_equal(G, g) and np.array_equal(B, b) and np.array_equal(S, s)
    want = eng.ss_get_structural(chains - 1)
    assert np.array_equal(model.state(chains - 1), want["state"].T)
    assert np.array_equal(model.state_variances(chains - 1), want["variances"])


def test_state_space_regression_with_an_ar_state_model():
    """bsts AddAr through the module: add_state(ArStateModel(lags)) after the trend"""
    import boom_amd._boom as boom
    rng = np.random.Generator(np.random.PCG64(3))
    T, p = 400, 3
    X = rng.standard_normal((T, p))
    u = np.zeros(T)
    for t in range(2, T):
        u[t] = 1.1 * u[t - 1] - 0.4 * u[t - 2] + 0.5 * rng.standard_normal()
    y = np.cumsum(0.02 * rng.standard_normal(T)) + u + X @ np.array([2.0, 0.0, -1.5]) \
        + 0.1 * rng.standard_normal(T)
    model = boom.StateSpaceRegressionModel(y, X, chains=4, seed=11)
    level = boom.LocalLevelStateModel(0.1)
    level.set_initial_state_mean(float(y[0]))
    level.set_initial_state_variance(4.0)
    level.set_prior(1.0, 0.05, 0.2)
    model.add_state(level)
    ar = boom.ArStateModel(2)
    ar.set_sigma(0.5)
    ar.set_initial_state_variance(2.0)
    ar.set_prior(1.0, 0.5)
    model.add_state(ar)
    assert model.state_dimension == 3
    slab = boom.MvnGivenScalarSigma(np.zeros(p), 0.01 * np.eye(p))
    sampler = boom.StateSpacePosteriorSampler(model, slab, boom.ChisqModel(1.0, 0.5),
                                              boom.VariableSelectionPrior(np.full(p, 0.5)))
    model.set_method(sampler)
    for _ in range(200):
        model.sample_posterior()
    phi = np.array([model.ar_phi(c) for c in range(4)])
    assert phi.shape == (4, 2)
    assert abs(phi[:, 0].mean() - 1.1) < 0.3 and abs(phi[:, 1].mean() + 0.4) < 0.3
    assert model.state(1).shape == (3, T)
    assert 0.05 < model.ar_sigsq(0) < 0.8


def test_any_state_list_on_the_pybind_module():
    """add_state in any order and number, a seasonal model with season_duration > 1 and a
    time_of_first_observation: boom.StateSpaceRegressionModel served from the look-ahead
    == the engine through the C-ABI, one ba_ss_sweep per iteration"""
    import boom_amd
    import boom_amd._boom as boom
    from cases import bsts_priors, general_data, general_spec
    T, p, seed, chains, niter = 100, 4, 8, 4, 9
    desc = [("seasonal", 4, 3, 2), ("trend",), ("ar", 2)]
    X, y, _, obs = general_data(T, p, 2, [(4, 3)], seed=14, missing_frac=0.04, ar_coef=[0.5])
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    model = boom.StateSpaceRegressionModel(y, X, [bool(o) for o in obs], chains=chains, seed=seed)
    b = blocks[0]
    seas = boom.SeasonalStateModel(4, 3)
    seas.set_time_of_first_observation(2)
    seas.set_sigsq(b["initial_sigma"][0] ** 2)
    seas.set_prior(b["df"][0], b["sigma_guess"][0], b["sigma_upper_limit"][0])
    seas.set_initial_state_mean(b["a0"])
    seas.set_initial_state_variance(b["P0"][0])
    b = blocks[1]
    trend = boom.LocalLinearTrendStateModel()
    trend.set_initial_sigma(b["initial_sigma"][0], b["initial_sigma"][1])
    for i in range(2):
        trend.set_prior(i, b["df"][i], b["sigma_guess"][i], b["sigma_upper_limit"][i])
    trend.set_initial_state_mean(b["a0"])
    trend.set_initial_state_variance(b["P0"])
    b = blocks[2]
    ar = boom.ArStateModel(2)
    ar.set_sigma(b["initial_sigma"][0])
    ar.set_prior(b["df"][0], b["sigma_guess"][0], b["sigma_upper_limit"][0])
    ar.set_initial_state_mean(b["a0"])
    ar.set_initial_state_variance(b["P0"][0])
    model.add_state(seas)
    model.add_state(trend)
    model.add_state(ar)
    assert model.number_of_state_models == 3 and model.state_dimension == 7
    sampler = boom.StateSpacePosteriorSampler(model, boom.MvnGivenScalarSigma(prior["b"], prior["ominv"]),
                                              boom.ChisqModel(prior["df"], prior["sigma_guess"]),
                                              boom.VariableSelectionPrior(prior["pi"]), sig_up)
    sampler.set_lookahead(4)
    model.set_method(sampler)
    eng = boom_amd.Engine(chains, seed=seed)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                   sigma_upper_limit=sig_up)
    eng.ss_set_state_models(blocks)
    eng.set_state(np.zeros(p, np.uint8))
    for it in range(niter):
        model.sample_posterior()
        eng.ss_sweep(1)
        G, B, S = model.chain_states()
        g, bb, s = eng.get_states()
        assert np.array_equal(G, g) and np.array_equal(B, bb) and np.array_equal(S, s), it
        assert np.array_equal(model.state(0), eng.ss_get_state_draw(0).T), it
        want = np.concatenate([eng.ss_get_state_model(0, k)["variances"] for k in range(3)])
        assert np.array_equal(model.state_variances(0), want), it
        assert np.array_equal(model.ar_phi(0), eng.ss_get_state_model(0, 2)["phi"]), it
    # a chain whose state path the look-ahead does not keep
    assert np.array_equal(model.state(chains - 1), eng.ss_get_state_draw(chains - 1).T)


def test_static_intercept_and_trig_on_the_pybind_module():
    """round 6: boom.StaticInterceptStateModel and boom.TrigStateModel handed to add_state (the
    module computes the rotations from period and frequencies as the reference does) == the
    engine through the C-ABI on the block list of tests/cases.py, bit for bit"""
    import boom_amd
    import boom_amd._boom as boom
    from cases import bsts_priors, general_data, general_spec
    T, p, seed, chains, niter = 96, 4, 21, 4, 9
    desc = [("intercept",), ("trig", 12.0, [1.0, 2.0, 3.0]), ("level",), ("semilocal",)]
    X, y, _, obs = general_data(T, p, 2, [], seed=15, missing_frac=0.03, trig=[(12.0, [1.0, 2.0])], intercept=4.0)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    model = boom.StateSpaceRegressionModel(y, X, [bool(o) for o in obs], chains=chains, seed=seed)
    b = blocks[0]
    icpt = boom.StaticInterceptStateModel()
    icpt.set_initial_state_mean(b["a0"][0])
    icpt.set_initial_state_variance(b["P0"][0])
    b = blocks[1]
    trig = boom.TrigStateModel(12.0, np.array([1.0, 2.0, 3.0]))
    trig.set_sigsq(b["initial_sigma"][0] ** 2)
    trig.set_prior(b["df"][0], b["sigma_guess"][0], b["sigma_upper_limit"][0])
    trig.set_initial_state_mean(b["a0"])
    trig.set_initial_state_variance(b["P0"])
    b = blocks[2]
    level = boom.LocalLevelStateModel(b["initial_sigma"][0])
    level.set_prior(b["df"][0], b["sigma_guess"][0], b["sigma_upper_limit"][0])
    level.set_initial_state_mean(b["a0"][0])
    level.set_initial_state_variance(b["P0"][0])
    b = blocks[3]
    sp = b["slope_priors"]
    semi = boom.SemilocalLinearTrendStateModel(boom.ZeroMeanGaussianModel(b["initial_sigma"][0]),
                                               boom.NonzeroMeanAr1Model(sp[4], sp[5], b["initial_sigma"][1]))
    semi.set_level_prior(b["df"][0], b["sigma_guess"][0], b["sigma_upper_limit"][0])
    semi.set_slope_prior(sp[0], sp[1], sp[2], sp[3], b["df"][1], b["sigma_guess"][1], b["sigma_upper_limit"][1],
                         bool(b["force_stationary"]), bool(b["force_positive"]))
    semi.set_initial_level_mean(b["a0"][0])
    semi.set_initial_slope_mean(b["a0"][1])
    semi.set_initial_level_sd(float(np.sqrt(b["P0"][0])))
    semi.set_initial_slope_sd(float(np.sqrt(b["P0"][1])))
    blocks[3]["P0"] = np.array([np.sqrt(b["P0"][0]) ** 2, np.sqrt(b["P0"][1]) ** 2, 0.0])   # (what the sd setters keep)
    for sm in (icpt, trig, level, semi):
        model.add_state(sm)
    assert model.number_of_state_models == 4 and model.state_dimension == 11 and trig.state_dimension == 6
    sampler = boom.StateSpacePosteriorSampler(model, boom.MvnGivenScalarSigma(prior["b"], prior["ominv"]),
                                              boom.ChisqModel(prior["df"], prior["sigma_guess"]),
                                              boom.VariableSelectionPrior(prior["pi"]), sig_up)
    sampler.set_lookahead(4)
    model.set_method(sampler)
    eng = boom_amd.Engine(chains, seed=seed)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                   sigma_upper_limit=sig_up)
    eng.ss_set_state_models(blocks)
    eng.set_state(np.zeros(p, np.uint8))
    for it in range(niter):
        model.sample_posterior()
        eng.ss_sweep(1)
        G, B, S = model.chain_states()
        g, bb, s = eng.get_states()
        assert np.array_equal(G, g) and np.array_equal(B, bb) and np.array_equal(S, s), it
        assert np.array_equal(model.state(0), eng.ss_get_state_draw(0).T), it
        want = np.concatenate([eng.ss_get_state_model(0, k)["variances"] for k in range(4)])
        assert len(want) == 4 and np.array_equal(model.state_variances(0), want), it
        assert np.array_equal(model.semilocal_slope(1), eng.ss_get_state_model(1, 3, suf=False)["phi"]), it
    st = model.state(chains - 1)
    assert np.all(st[0] == st[0, 0])     # the intercept is one number per draw


def test_poisson_spike_slab_on_the_pybind_module():
    """boom.PoissonRegressionModel + PoissonRegressionSpikeSlabSampler: the draws of the
    engine through the C-ABI (ba_poisson_*), the mixture table handed over as data"""
    import boom_amd
    import boom_amd._boom as boom
    from test_oracle_golden import _golden_mix, load
    g = load("poisson_exposure")
    X, y, ex, mix = g["X"], g["y"], g["exposure"], _golden_mix(g)
    p = X.shape[1]
    chains, seed, niter = 4, 23, 8
    model = boom.PoissonRegressionModel(X, y, ex, chains=chains, seed=seed)
    model.set_mixture_table([int(c) for c in mix["counts"]], [int(c) for c in mix["ncomp"]], mix["mu"],
                            mix["sigma"], mix["weight"], int(mix["largest_index"]))
    model.drop_all()
    for j in np.flatnonzero(g["init_gamma"]):
        model.add(int(j))
    sampler = boom.PoissonRegressionSpikeSlabSampler(model, boom.MvnModel(g["mu"], g["prec"], True),
                                                     boom.VariableSelectionPrior(g["pi"]))
    model.set_method(sampler)
    eng = boom_amd.Engine(chains, seed=seed)
    eng.poisson_set_data(X, y, ex, mix)
    eng.sss_set_slab(g["mu"], g["prec"], scales_with_sigsq=False)
    eng.set_spike(g["pi"])
    eng.set_state(g["init_gamma"])
    for it in range(niter):
        model.sample_posterior()
        eng.poisson_sweep(1)
        assert np.array_equal(np.asarray(model.inc, np.uint8), eng.get_state(0)[0]), it
        assert np.array_equal(model.Beta, eng.get_state(0)[1]), it
    G, B = model.chain_states()
    assert np.array_equal(G, eng.get_states()[0]) and np.array_equal(B, eng.get_states()[1])
