"""bsts structural time series on the device: regression +
local level / local linear trend + seasonal state, against the CPU oracle
(itself pinned on the reference: tests/golden/ssm_*.npz), through the C-ABI.

Bars as for the local-level path: inclusion indicators bit-exact; beta, sigma^2,
the state models' variances and the state draw within 1e-8 relative.
"""
import numpy as np
import pytest

from cases import bsts_priors, structural_data, structural_spec
from oracle_lib import ssvs_options

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def make_engine(chains, seed, y, X, obs, prior, spec, sig_up, g0):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], sigma_upper_limit=sig_up)
    m0 = trend_dim(spec)
    eng.ss_set_structural(spec["trend"], spec["nseasons"], spec["var_df"],
                          spec["var_sigma_guess"], spec["var_sigma_upper_limit"],
                          spec["var_initial_sigma"], spec["initial_state_mean"][:m0],
                          spec["initial_state_variance"][:m0])
    ar = spec.get("ar")
    if ar:
        m0 = trend_dim(spec)
        eng.ss_add_ar(ar["lags"], ar["df"], ar["sigma_guess"], ar["sigma_upper_limit"],
                      ar["initial_sigma"], ar["initial_phi"], spec["initial_state_mean"][m0:],
                      spec["initial_state_variance"][m0:])
    eng.set_state(g0)
    return eng


def trend_dim(spec):
    return spec["trend"] + (spec["nseasons"] - 1 if spec["nseasons"] > 0 else 0)


def used(trend, ns):
    return [0] + ([1] if trend == 2 else []) + ([2] if ns > 0 else [])


@pytest.mark.parametrize("trend,nseasons,T,missing",
                         [(1, 0, 150, 0.0), (2, 0, 150, 0.0), (1, 7, 200, 0.0),
                          (2, 4, 150, 0.05), (2, 12, 300, 0.0), (2, 15, 130, 0.03),
                          (1, 2, 65, 0.0), (2, 7, 700, 0.0), (2, 4, 3, 0.0), (1, 3, 2, 0.0),
                          (2, 0, 64, 0.0), (2, 5, 128, 0.02), (1, 16, 70, 0.0), (2, 12, 2100, 0.02),
                          (1, 12, 13, 0.0)])
def test_structural_sweeps_match_oracle(oracle, trend, nseasons, T, missing):
    p, chains, seed, nsw = 6, 5, 29, 12
    X, y, _, obs = structural_data(T, p, 2, nseasons, seed=3 + nseasons, missing_frac=missing)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, trend, nseasons)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, spec, sig_up, g0)
    check = [0, chains - 1]
    ora = {c: oracle.ssm_run(y, X, obs, prior, opts, spec, ("philox", seed, c), g0, nsw)
           for c in check}
    idx = used(trend, nseasons)
    for s in range(nsw):
        eng.ss_sweep(1)
        gam, beta, sig = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            tag = (trend, nseasons, c, s)
            assert np.array_equal(gam[c], o["gamma"][s]), tag
            assert relerr(beta[c], o["beta"][s]) < RTOL, tag
            assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], tag
            st = eng.ss_get_structural(c)
            assert relerr(st["variances"][idx], o["variances"][s][idx], 1e-300) < RTOL, tag
            scale = np.abs(o["state"][s]).max()
            assert np.max(np.abs(st["state"] - o["state"][s])) < 1e-8 * scale, tag


def test_structural_many_sweeps_in_one_call_and_shapes(oracle):
    """one ba_ss_sweep(n) call == n calls of one; a seasonal pattern and a trend
    are recovered (size-independent properties on a longer series)"""
    T, p, ns = 600, 8, 7
    X, y, btrue, obs = structural_data(T, p, 3, ns, seed=11)
    prior, _, sig_up = bsts_priors(X, y, 3)
    spec = structural_spec(y, 2, ns)
    g0 = np.zeros(p, np.uint8)
    a = make_engine(16, 3, y, X, obs, prior, spec, sig_up, g0)
    b = make_engine(16, 3, y, X, obs, prior, spec, sig_up, g0)
    a.ss_sweep(30)
    for _ in range(30):
        b.ss_sweep(1)
    ga, ba_, sa = a.get_states()
    gb, bb, sb = b.get_states()
    assert np.array_equal(ga, gb) and np.array_equal(ba_, bb) and np.array_equal(sa, sb)
    assert ga[:, :3].all()
    assert np.max(np.abs(ba_[:, :3].mean(axis=0) - btrue[:3])) < 0.3
    st = a.ss_get_structural(0)
    fitted = st["state"][:, 0] + st["state"][:, 2] + X @ ba_[0]
    assert np.sqrt(np.mean((fitted - y) ** 2)) < 0.5
    assert np.all(st["variances"] > 0) and np.all(st["suf_n"] == T - 1)


@pytest.mark.parametrize("trend,nseasons", [(1, 0), (2, 0), (1, 7), (2, 12)])
def test_structural_forecast_matches_oracle(oracle, trend, nseasons):
    """simulate_forecast for every chain's current draw: the state advances by its
    transition plus state errors, the observation adds noise and x'beta; normals in
    the reference's order on the chain's forecast stream"""
    T, p, chains, seed, h = 150, 6, 5, 23, 30
    X, y, _, obs = structural_data(T, p, 2, nseasons, seed=3 + nseasons)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, trend, nseasons)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, spec, sig_up, g0)
    eng.ss_sweep(15)
    newX = np.random.Generator(np.random.PCG64(8)).standard_normal((h, p))
    f1 = eng.ss_forecast(newX)
    f2 = eng.ss_forecast(newX)          # the streams continue: a second, different draw
    gam, beta, sig = eng.get_states()
    assert f1.shape == (chains, h) and not np.array_equal(f1, f2)
    for c in range(chains):
        st = eng.ss_get_structural(c)
        rng = oracle.rng_philox(seed, chain=c, stream=5)
        want1 = oracle.ssm_forecast(rng, newX, beta[c], sig[c], trend, nseasons, st["variances"],
                                    st["state"][-1])
        want2 = oracle.ssm_forecast(rng, newX, beta[c], sig[c], trend, nseasons, st["variances"],
                                    st["state"][-1])
        assert np.max(np.abs(f1[c] - want1)) < 1e-9 * np.abs(want1).max()
        assert np.max(np.abs(f2[c] - want2)) < 1e-9 * np.abs(want2).max()


# ----------------------------------------------------------------- ArStateModel block
AR_CASES = [  # trend, nseasons, T, missing, coefficients of the data's autoregression
    (1, 0, 150, 0.0, [0.8]),
    (2, 4, 160, 0.04, [1.2, -0.4]),
    (1, 0, 220, 0.0, [0.9, 0.3, -0.35]),
    (2, 7, 130, 0.0, [0.5, 0.2, 0.1, -0.2, 0.1, 0.05, -0.1, 0.05]),   # m = 2 + 6 + 8 = 16
    # short series: the proposals are rarely stationary, the coefficients come one at a
    # time from two-sided truncated normals (both interior samplers and the Tn2Sampler)
    (1, 0, 3, 0.0, [0.5, -0.2]),
    (1, 0, 2, 0.0, [0.5]),
    (1, 0, 6, 0.0, [0.5, -0.2]),
    (2, 3, 65, 0.0, [1.4, -0.6]),
]


@pytest.mark.parametrize("trend,nseasons,T,missing,coef", AR_CASES)
def test_structural_ar_sweeps_match_oracle(oracle, trend, nseasons, T, missing, coef):
    """regression + trend [+ seasonal] + ArStateModel(lags): the state draw, the
    autoregression coefficients (multivariate proposals with the stationarity check),
    its error variance and its sufficient statistics, chain by chain against the oracle"""
    p, chains, seed, nsw = 6, 5, 31, 12
    X, y, _, obs = structural_data(T, p, 2, nseasons, seed=3 + nseasons, missing_frac=missing,
                                   ar_coef=coef)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, trend, nseasons, ar_lags=len(coef))
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, spec, sig_up, g0)
    check = [0, chains - 1]
    ora = {c: oracle.ssm_run(y, X, obs, prior, opts, spec, ("philox", seed, c), g0, nsw)
           for c in check}
    idx = used(trend, nseasons)
    for s in range(nsw):
        eng.ss_sweep(1)
        gam, beta, sig = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            tag = (trend, nseasons, len(coef), c, s)
            assert np.array_equal(gam[c], o["gamma"][s]), tag
            assert relerr(beta[c], o["beta"][s]) < RTOL, tag
            assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], tag
            st = eng.ss_get_structural(c)
            assert relerr(st["variances"][idx], o["variances"][s][idx], 1e-300) < RTOL, tag
            scale = np.abs(o["state"][s]).max()
            assert np.max(np.abs(st["state"] - o["state"][s])) < 1e-8 * scale, tag
            ar = eng.ss_get_ar(c)
            assert relerr(ar["phi"], o["ar_phi"][s], 1e-2) < 1e-7, tag
            assert abs(ar["sigsq"] - o["ar_sigsq"][s]) < 1e-7 * o["ar_sigsq"][s], tag
            assert ar["n"] == T - 1
            # the sufficient statistics are those of the state draw
            a0 = trend_dim(spec)
            blk = st["state"][:, a0:]
            assert np.allclose(ar["xtx"], blk[:-1].T @ blk[:-1], rtol=1e-9, atol=1e-12), tag
            assert np.allclose(ar["xty"], blk[:-1].T @ blk[1:, 0], rtol=1e-9, atol=1e-12), tag


def test_structural_ar_many_sweeps_in_one_call_and_recovery(oracle):
    """one ba_ss_sweep(n) == n calls of one; every accepted coefficient vector is
    stationary; the autoregression of the data is found"""
    T, p, coef = 500, 6, [1.2, -0.4]
    X, y, _, obs = structural_data(T, p, 2, 0, seed=9, ar_coef=coef)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, 1, 0, ar_lags=2)
    g0 = np.zeros(p, np.uint8)
    a = make_engine(32, 5, y, X, obs, prior, spec, sig_up, g0)
    b = make_engine(32, 5, y, X, obs, prior, spec, sig_up, g0)
    a.ss_sweep(40)
    for _ in range(40):
        b.ss_sweep(1)
    assert all(np.array_equal(u, v) for u, v in zip(a.get_states(), b.get_states()))
    phis = np.array([a.ss_get_ar(c)["phi"] for c in range(32)])
    assert np.array_equal(phis, np.array([b.ss_get_ar(c)["phi"] for c in range(32)]))
    for ph in phis:
        roots = np.roots(np.r_[-ph[::-1], 1.0])
        assert np.all(np.abs(roots) > 1.0)


def test_structural_ar_forecast_matches_oracle(oracle):
    T, p, chains, seed, h = 150, 6, 4, 23, 25
    X, y, _, obs = structural_data(T, p, 2, 4, seed=7, ar_coef=[0.7, -0.2])
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, 2, 4, ar_lags=2)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, spec, sig_up, g0)
    eng.ss_sweep(10)
    newX = np.random.Generator(np.random.PCG64(8)).standard_normal((h, p))
    f1 = eng.ss_forecast(newX)
    gam, beta, sig = eng.get_states()
    for c in range(chains):
        st, ar = eng.ss_get_structural(c), eng.ss_get_ar(c)
        rng = oracle.rng_philox(seed, chain=c, stream=5)
        want = oracle.ssm_forecast(rng, newX, beta[c], sig[c], 2, 4, st["variances"],
                                   st["state"][-1], ar_phi=ar["phi"], ar_sigsq=ar["sigsq"])
        assert np.max(np.abs(f1[c] - want)) < 1e-9 * np.abs(want).max(), c


def test_structural_ar_rejects_bad_arguments():
    import boom_amd
    T, p = 50, 3
    X, y, _, obs = structural_data(T, p, 1, 0, seed=1)
    eng = boom_amd.Engine(2, seed=1)
    eng.ss_set_data(y, X, obs)
    with pytest.raises(boom_amd.BoomAmdError):            # before ba_ss_set_structural
        eng.ss_add_ar(1, 0.01, 0.1, 1.0, 1.0, None, np.zeros(1), np.ones(1))
    spec = structural_spec(y, 2, 61)
    eng.ss_set_structural(2, 61, spec["var_df"], spec["var_sigma_guess"],
                          spec["var_sigma_upper_limit"], spec["var_initial_sigma"],
                          spec["initial_state_mean"], spec["initial_state_variance"])
    with pytest.raises(boom_amd.BoomAmdError):            # 62 + 4 > 64
        eng.ss_add_ar(4, 0.01, 0.1, 1.0, 1.0, None, np.zeros(4), np.ones(4))
    with pytest.raises(boom_amd.BoomAmdError):            # more than 16 lags
        eng.ss_add_ar(17, 0.01, 0.1, 1.0, 1.0, None, np.zeros(17), np.ones(17))
    spec = structural_spec(y, 2, 12)
    eng.ss_set_structural(2, 12, spec["var_df"], spec["var_sigma_guess"],
                          spec["var_sigma_upper_limit"], spec["var_initial_sigma"],
                          spec["initial_state_mean"], spec["initial_state_variance"])
    with pytest.raises(boom_amd.BoomAmdError):            # a unit root
        eng.ss_add_ar(2, 0.01, 0.1, 1.0, 1.0, np.array([1.5, -0.5]), np.zeros(2), np.ones(2))
    with pytest.raises(boom_amd.BoomAmdError):            # variance 0
        eng.ss_add_ar(2, 0.01, 0.1, 1.0, 1.0, None, np.zeros(2), np.array([1.0, 0.0]))
    eng.ss_add_ar(2, 0.01, 0.1, 1.0, 1.0, np.array([1.2, -0.4]), np.zeros(2), np.ones(2))
    with pytest.raises(boom_amd.BoomAmdError):            # only one block
        eng.ss_add_ar(1, 0.01, 0.1, 1.0, 1.0, None, np.zeros(1), np.ones(1))
