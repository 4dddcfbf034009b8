"""bsts structural time series on the device, the general form: ANY
list of state models in any order -- seasonal blocks with season_duration > 1 and a
time_of_first_observation, two seasonal blocks, models without a trend block, two
autoregression blocks, state dimension up to 64 -- against the CPU oracle (itself pinned on
the compiled reference: tests/golden/ssg_*.npz), through the C-ABI
(ba_ss_add_state_model ...).

Bars as for the local-level path: inclusion indicators bit-exact; beta, sigma^2, the state
models' variances, coefficients and sufficient statistics and the state draw within 1e-8
relative.
"""
import numpy as np
import pytest

from cases import blocks_of, bsts_priors, general_data, general_spec
from oracle_lib import ssvs_options
from test_oracle_golden import GENERAL_GOLDENS, load, opts_of, prior_of

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_state_models(blocks)
    eng.set_state(g0)
    return eng


def compare(eng, oracle, y, X, obs, prior, opts, blocks, seed, chains, g0, nsw, check=None):
    check = check or [0, chains - 1]
    ora = {c: oracle.ssg_run(y, X, obs, prior, opts, blocks, ("philox", seed, c), g0, nsw)
           for c in check}
    for s in range(nsw):
        eng.ss_sweep(1)
        gam, beta, sig = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            tag = (c, s)
            assert np.array_equal(gam[c], o["gamma"][s]), tag
            assert relerr(beta[c], o["beta"][s]) < RTOL, tag
            assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], tag
            for b, blk in enumerate(blocks):
                sm = eng.ss_get_state_model(c, b)
                nv = len(sm["variances"])
                if nv == 0:      # (a static intercept has no parameter)
                    continue
                assert relerr(sm["variances"], o["variances"][s, b, :nv], 1e-300) < RTOL, tag + (b,)
                if blk["kind"] == 4:
                    assert relerr(sm["phi"], o["phi"][s, b, :blk["lags"]], 1e-6) < RTOL, tag + (b,)
                if blk["kind"] == 7:     # (a semilocal trend's phi and mu)
                    assert relerr(sm["phi"], o["phi"][s, b, :2], 1e-6) < RTOL, tag + (b,)
            st = eng.ss_get_state_draw(c)
            scale = np.abs(o["state"][s]).max()
            assert np.max(np.abs(st - o["state"][s])) < 1e-8 * scale, tag
    # the sufficient statistics the last impute_state left behind
    for c in check:
        o = ora[c]
        for b, blk in enumerate(blocks):
            sm = eng.ss_get_state_model(c, b)
            nv = len(sm["variances"])
            if blk["kind"] == 4:
                a = o["ar_suf"][b]
                sc = np.abs(a["xtx"]).max()
                assert np.max(np.abs(sm["xtx"] - a["xtx"])) < 1e-8 * sc
                assert np.max(np.abs(sm["xty"] - a["xty"])) < 1e-8 * sc
                assert abs(sm["yty"] - a["yty"]) < 1e-8 * sc and sm["n"] == a["n"]
            elif blk["kind"] == 7:   # (the level's statistics; the slope's Ar1Suf has n = T)
                assert sm["suf_n"][0] == o["suf_n"][b, 0] and sm["ar1_suf"][3] == len(y), (c, b)
                assert relerr(sm["suf_ss"][:1], o["suf_ss"][b, :1], 1e-300) < RTOL, (c, b)
            elif nv > 0:
                assert np.array_equal(sm["suf_n"], o["suf_n"][b, :nv]), (c, b)
                assert relerr(sm["suf_ss"], o["suf_ss"][b, :nv], 1e-300) < RTOL, (c, b)


@pytest.mark.parametrize("name", GENERAL_GOLDENS)
def test_general_state_lists_match_oracle(oracle, name):
    """the data and specifications of the reference goldens (seasonal-only,
    autoregression-only, weekly + a 4-season cycle of duration 7, a seasonal block ahead of
    the level with missing observations, a time_of_first_observation, two autoregression
    blocks, a local level beside a local linear trend, 52 seasons of duration 7: m = 53)"""
    g = load(name)
    blocks = blocks_of(g)
    obs = g["observed"]
    obs = None if obs.all() else obs
    prior, opts = prior_of(g), opts_of(g)
    p = g["X"].shape[1]
    g0 = np.zeros(p, np.uint8)
    chains, seed = 5, 61
    nsw = 6 if name == "ssg_big52" else 10
    eng = make_engine(chains, seed, g["y"], g["X"], obs, prior, blocks, opts["sigma_upper_limit"], g0)
    compare(eng, oracle, g["y"], g["X"], obs, prior, opts, blocks, seed, chains, g0, nsw)


@pytest.mark.parametrize("desc,T,missing", [
    # the state dimension's limit: 2 + 51 + 6 + 1 + 4 = 64, eight variance parameters
    ([("trend",), ("seasonal", 52, 7), ("seasonal", 7, 1), ("level",), ("ar", 4)], 380, 0.02),
    # the passes' block length does not divide T (m <= 16: 64, m <= 32: 32, else 16)
    ([("seasonal", 12, 1), ("trend",), ("seasonal", 5, 3, 1)], 97, 0.0),
    ([("level",), ("seasonal", 24, 2)], 131, 0.04),
    # eight state models
    ([("level",), ("ar", 1), ("seasonal", 3, 1), ("trend",), ("seasonal", 2, 5), ("ar", 2),
      ("seasonal", 4, 2, 3), ("level",)], 90, 0.0),
    # series of two and three points
    ([("seasonal", 3, 2), ("trend",)], 3, 0.0),
    ([("ar", 1), ("seasonal", 4, 1)], 2, 0.0),
])
def test_general_shapes_match_oracle(oracle, desc, T, missing):
    p, chains, seed, nsw = 5, 4, 17, 8
    seas = [(b[1], b[2]) for b in desc if b[0] == "seasonal"]
    X, y, _, obs = general_data(T, p, 2, seas[:2], seed=5 + T, missing_frac=missing,
                                ar_coef=[0.5] if any(b[0] == "ar" for b in desc) else None)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    compare(eng, oracle, y, X, obs, prior, opts, blocks, seed, chains, g0, nsw)


def test_general_form_equals_the_template():
    """[trend, seasonal(n, 1), ar] through ba_ss_add_state_model == ba_ss_set_structural +
    ba_ss_add_ar, bit for bit; one call of n sweeps == n calls"""
    import boom_amd
    from cases import structural_data, structural_spec
    T, p, chains, seed = 140, 6, 6, 9
    X, y, _, obs = structural_data(T, p, 2, 6, seed=4, missing_frac=0.03, ar_coef=[0.4])
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, 2, 6, ar_lags=1)
    g0 = np.zeros(p, np.uint8)
    a = boom_amd.Engine(chains, seed=seed)
    a.ss_set_data(y, X, obs)
    a.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                 sigma_upper_limit=sig_up)
    a.ss_set_structural(2, 6, spec["var_df"], spec["var_sigma_guess"],
                        spec["var_sigma_upper_limit"], spec["var_initial_sigma"],
                        spec["initial_state_mean"][:7], spec["initial_state_variance"][:7])
    ar = spec["ar"]
    a.ss_add_ar(1, ar["df"], ar["sigma_guess"], ar["sigma_upper_limit"], ar["initial_sigma"],
                ar["initial_phi"], spec["initial_state_mean"][7:], spec["initial_state_variance"][7:])
    a.set_state(g0)
    blocks = general_spec(y, [("trend",), ("seasonal", 6, 1), ("ar", 1)])
    b = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    a.ss_sweep(9)
    for _ in range(9):
        b.ss_sweep(1)
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)
    sa = a.ss_get_structural(2)
    assert np.array_equal(sa["state"], b.ss_get_state_draw(2))
    assert np.array_equal(sa["variances"][:2], b.ss_get_state_model(2, 0)["variances"])
    assert sa["variances"][2] == b.ss_get_state_model(2, 1)["variances"][0]
    assert np.array_equal(a.ss_get_ar(2)["phi"], b.ss_get_state_model(2, 2)["phi"])


@pytest.mark.parametrize("desc,T", [
    ([("level",)], 130), ([("trend",), ("seasonal", 12, 1)], 300),
    ([("level",), ("seasonal", 7, 1), ("ar", 2)], 97), ([("trend",), ("ar", 3)], 150)])
def test_general_kernel_equals_the_shape_specialised_one(desc, T):
    """block lists of the template's shape run a kernel compiled for the shape
    (ssm_template_kernel.hip); ba_ss_set_tuning(0) sends them through the general kernel:
    the same chains (inclusion indicators identical, the rest within 1e-9 -- the two
    kernels sum in different orders)"""
    p, chains, seed = 5, 6, 13
    seas = [(b[1], b[2]) for b in desc if b[0] == "seasonal"]
    X, y, _, obs = general_data(T, p, 2, seas, seed=T, missing_frac=0.03,
                                ar_coef=[0.5] if any(b[0] == "ar" for b in desc) else None)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    g0 = np.zeros(p, np.uint8)
    a = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    b = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    b.ss_set_tuning(use_template_kernel=False)
    for s in range(8):
        a.ss_sweep(1)
        b.ss_sweep(1)
        ga, ba_, sa = a.get_states()
        gb, bb, sb = b.get_states()
        assert np.array_equal(ga, gb), s
        assert relerr(ba_, bb) < 1e-9 and relerr(sa, sb, 1e-300) < 1e-9, s
        for c in (0, chains - 1):
            u, v = a.ss_get_state_draw(c), b.ss_get_state_draw(c)
            assert np.max(np.abs(u - v)) < 1e-9 * np.abs(v).max(), (s, c)
            for k in range(len(blocks)):
                mu, mv = a.ss_get_state_model(c, k), b.ss_get_state_model(c, k)
                assert relerr(mu["variances"], mv["variances"], 1e-300) < 1e-9, (s, c, k)
                assert relerr(mu["suf_ss"], mv["suf_ss"], 1e-300) < 1e-9, (s, c, k)


@pytest.mark.parametrize("key", ["a", "b", "c"])
def test_general_forecast_matches_oracle(oracle, key):
    """simulate_forecast with seasonal blocks of duration > 1 (the reference simulates
    forecast step i with the matrices of time T - 2 + i): the device's forecast of every
    chain's current draw against the oracle's on the chain's forecast stream"""
    g = load("kat_general_forecast")
    desc = {"a": [("level",), ("seasonal", 4, 3)],
            "b": [("seasonal", 3, 5, 1), ("trend",), ("ar", 2)],
            "c": [("trend",), ("seasonal", 7, 1), ("seasonal", 4, 7)]}[key]
    T, p, chains, seed, h = int(g[key + "_T"]), 6, 4, 31, 25
    seas = [(b[1], b[2]) for b in desc if b[0] == "seasonal"]
    X, y, _, obs = general_data(T, p, 2, seas, seed=77)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    eng.ss_sweep(8)
    newX = np.random.Generator(np.random.PCG64(8)).standard_normal((h, p))
    fc = eng.ss_forecast(newX)
    gam, beta, sig = eng.get_states()
    for c in (0, chains - 1):
        st = eng.ss_get_state_draw(c)
        sg = np.zeros((len(blocks), 2))
        ph = np.zeros((len(blocks), 16))
        for b, blk in enumerate(blocks):
            sm = eng.ss_get_state_model(c, b)
            sg[b, :len(sm["variances"])] = sm["variances"]
            if blk["kind"] == 4:
                ph[b, :blk["lags"]] = sm["phi"]
        want = oracle.ssg_forecast(oracle.rng_philox(seed, c, 5), T, newX, beta[c], sig[c], blocks,
                                   sg, ph, st[-1])
        assert np.max(np.abs(fc[c] - want)) < 1e-8 * max(1.0, np.abs(want).max()), (key, c)


def test_general_argument_errors():
    import boom_amd
    T, p = 40, 3
    X, y, _, _ = general_data(T, p, 1, [], seed=1)
    eng = boom_amd.Engine(2, seed=1)
    eng.ss_set_data(y, X, None)
    ok = general_spec(y, [("level",)])[0]

    def bad(**kw):
        b = dict(ok)
        b.update(kw)
        with pytest.raises(boom_amd.BoomAmdError):
            eng.ss_set_state_models([b])
    bad(kind=8)
    sl = dict(kind=7, df=np.ones(2), sigma_guess=np.ones(2), sigma_upper_limit=np.full(2, np.inf),
              initial_sigma=np.ones(2), a0=np.zeros(3), P0=np.array([1.0, 1.0, 0.0]),
              slope_priors=np.array([0.0, 1.0, 0.0, 1.0, 0.0, 0.0]))
    bad(**dict(sl, force_stationary=0, force_positive=1))            # the one-sided truncation is not built
    bad(**dict(sl, slope_priors=np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0])))   # a prior sd of 0
    bad(**dict(sl, initial_sigma=np.array([1.0, 0.0])))
    bad(**dict(sl, P0=np.array([1.0, 0.0, 0.0])))
    bad(kind=6, rotations=np.zeros(0), a0=np.zeros(0), P0=np.zeros(0))              # no frequency
    bad(kind=6, rotations=np.tile([1.0, 0.0], 33), a0=np.zeros(66), P0=np.ones(66))  # 66 components
    bad(kind=5, df=np.zeros(0), sigma_guess=np.zeros(0), sigma_upper_limit=np.zeros(0),
        initial_sigma=np.zeros(0), a0=np.zeros(1), P0=np.array([-1.0]))             # "must be non-negative"
    bad(kind=3, nseasons=1, duration=1, a0=np.zeros(0), P0=np.zeros(0))
    bad(kind=3, nseasons=4, duration=0, a0=np.zeros(3), P0=np.ones(3))
    bad(kind=4, lags=17, a0=np.zeros(17), P0=np.ones(17), initial_phi=np.zeros(17))
    bad(kind=4, lags=1, a0=np.zeros(1), P0=np.ones(1), initial_phi=np.array([1.0]))
    bad(kind=2, df=np.ones(2), sigma_guess=np.ones(2), sigma_upper_limit=np.array([1.0, -1.0]),
        initial_sigma=np.ones(2), a0=np.zeros(2), P0=np.ones(2))
    bad(kind=3, nseasons=5, duration=1, a0=np.zeros(4), P0=np.array([1.0, 1.0, 0.0, 1.0]))
    # 66 components
    with pytest.raises(boom_amd.BoomAmdError):
        eng.ss_set_state_models(general_spec(y, [("seasonal", 60, 1), ("seasonal", 8, 1)]))
    # nine models
    with pytest.raises(boom_amd.BoomAmdError):
        eng.ss_set_state_models(general_spec(y, [("level",)] * 9))
    # a failed specification leaves nothing half-built
    with pytest.raises(boom_amd.BoomAmdError):
        eng.ss_sweep(1)
    eng.set_priors(np.zeros(p), np.eye(p), np.full(p, 0.5), 1.0, 1.0)
    eng.ss_set_state_models(general_spec(y, [("seasonal", 4, 2)]))
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_sweep(2)
    assert eng.ss_get_state_draw(0).shape == (T, 3)
