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
    eng.ss_set_structural(spec["trend"], spec["nseasons"], spec["var_df"],
                          spec["var_sigma_guess"], spec["var_sigma_upper_limit"],
                          spec["var_initial_sigma"], spec["initial_state_mean"],
                          spec["initial_state_variance"])
    eng.set_state(g0)
    return eng


def used(trend, ns):
    return [0] + ([1] if trend == 2 else []) + ([2] if ns > 0 else [])


@pytest.mark.parametrize("trend,nseasons,T,missing",
                         [(1, 0, 150, 0.0), (2, 0, 150, 0.0), (1, 7, 200, 0.0),
                          (2, 4, 150, 0.05), (2, 12, 300, 0.0), (2, 15, 130, 0.03),
                          (1, 2, 65, 0.0), (2, 7, 700, 0.0), (2, 4, 3, 0.0), (1, 3, 2, 0.0),
                          (2, 0, 64, 0.0), (2, 5, 128, 0.02)])
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
