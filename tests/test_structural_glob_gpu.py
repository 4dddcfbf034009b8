"""f2 along its glob on the device (VERDICT r5 task 8): StaticInterceptStateModel
(StaticInterceptStateModel.hpp:35-131) and TrigStateModel (TrigStateModel.cpp:130-223) as
blocks of the general structural kernel (ssm_kernel.hip), through the C-ABI
(ba_ss_add_state_model kinds 5 and 6), against the CPU oracle -- itself pinned on the
compiled reference by tests/golden/ssq_*.npz and kat_glob_forecast.npz.

Bars as for every state list: inclusion indicators bit-exact; beta, sigma^2, the state
models' variances and sufficient statistics and the state draw within 1e-8 relative."""
import numpy as np
import pytest

from cases import blocks_of, bsts_priors, general_data, general_spec
from oracle_lib import ssvs_options
from test_oracle_golden import GLOB_GOLDENS, load, opts_of, prior_of
from test_structural_general_gpu import compare, make_engine, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GLOB_GOLDENS)
def test_static_intercept_and_trig_lists_match_oracle(oracle, name):
    """the data and specifications of the six reference goldens: intercept + AR(2), a seasonal
    block ahead of the intercept with missing observations, trig alone, trend + three
    frequencies, trig ahead of level + weekly seasonal, intercept + two trig blocks"""
    g = load(name)
    blocks = blocks_of(g)
    obs = g["observed"]
    obs = None if obs.all() else obs
    prior, opts = prior_of(g), opts_of(g)
    p = g["X"].shape[1]
    g0 = np.zeros(p, np.uint8)
    chains, seed = 5, 71
    eng = make_engine(chains, seed, g["y"], g["X"], obs, prior, blocks, opts["sigma_upper_limit"], g0)
    compare(eng, oracle, g["y"], g["X"], obs, prior, opts, blocks, seed, chains, g0, 10)
    eng.close()


@pytest.mark.parametrize("desc,T,missing", [
    # sixteen frequencies: 32 components in one block (the non-SMALL kernel, leading dimension 33)
    ([("trig", 365.25, list(range(1, 17)))], 120, 0.0),
    # m = 1 + 2 + 20 + 6 + 11 + 1 = 41, six blocks, the intercept last; T not a multiple of the passes' block
    ([("level",), ("trend",), ("trig", 52.0, list(range(1, 11))), ("seasonal", 7, 1), ("seasonal", 12, 3),
      ("intercept",)], 101, 0.03),
    # the intercept alone with the regression (a series of three points too)
    ([("intercept",)], 60, 0.05),
    ([("intercept",), ("trig", 4.0, [1.0])], 3, 0.0),
    # a trig block with a frequency of half the period (sin = 1.2e-16: the second component is all but static)
    ([("trig", 6.0, [3.0, 1.0]), ("ar", 2)], 90, 0.0),
    # semilocal linear trend: with an autoregression block (both take coefficient slots), two of them,
    # in a list of m = 3 + 11 + 6 + 1 = 21, and on a series of three points
    ([("semilocal",), ("ar", 2)], 110, 0.02),
    ([("semilocal", 1, 1), ("seasonal", 4, 2), ("semilocal", 0, 0)], 95, 0.0),
    ([("seasonal", 12, 1), ("semilocal",), ("trig", 7.0, [1.0, 2.0, 3.0]), ("intercept",)], 130, 0.03),
    ([("semilocal",)], 3, 0.0),
])
def test_glob_shapes_match_oracle(oracle, desc, T, missing):
    p, chains, seed, nsw = 5, 4, 19, 8
    seas = [(b[1], b[2]) for b in desc if b[0] == "seasonal"]
    trig = [(b[1], b[2][:2]) for b in desc if b[0] == "trig"]
    X, y, _, obs = general_data(T, p, 2, seas[:2], seed=7 + T, missing_frac=missing, trig=trig, intercept=1.0,
                                ar_coef=[0.5] if any(b[0] == "ar" for b in desc) else None,
                                level=any(b[0] in ("level", "trend", "semilocal") for b in desc))
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    compare(eng, oracle, y, X, obs, prior, opts, blocks, seed, chains, g0, nsw)
    eng.close()


def test_static_intercept_is_no_local_level_of_the_stream_families(oracle):
    """a static intercept has no sampler: a local level AFTER it is still the first of its family
    (sampler id 1), as in the oracle's bo_ssm_block_stream_id"""
    T, p, chains, seed = 70, 4, 3, 5
    X, y, _, obs = general_data(T, p, 2, [], seed=3, intercept=2.0)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, [("intercept",), ("level",), ("level",)])
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    compare(eng, oracle, y, X, obs, prior, opts, blocks, seed, chains, g0, 6)
    eng.close()


@pytest.mark.parametrize("key", ["a", "b", "c"])
def test_glob_forecast_matches_oracle(oracle, key):
    """simulate_forecast with a static intercept (no error term drawn) and trig blocks (2 nfreq
    draws a step on the rotated state): every chain's forecast of its current draw against the
    oracle's on the chain's forecast stream"""
    g = load("kat_glob_forecast")
    desc = {"a": [("intercept",), ("trig", 12.0, [1.0, 2.0])],
            "b": [("trig", 7.0, [1.0, 2.0, 3.0]), ("trend",), ("seasonal", 3, 5, 1)],
            "c": [("seasonal", 4, 2), ("semilocal",)]}[key]
    T, p, chains, seed, h = int(g[key + "_T"]), 6, 4, 33, 20
    X, y, _, obs = general_data(T, p, 2, [], seed=79, trig=[(12.0 if key == "a" else 7.0, [1.0])])
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
        fb = [dict(b) for b in blocks]
        for b in range(len(blocks)):
            sm = eng.ss_get_state_model(c, b)
            sg[b, :len(sm["variances"])] = sm["variances"]
            if fb[b]["kind"] == 7:     # (the chain's phi and mu go in where the oracle's model is built)
                fb[b]["slope_priors"] = np.array(list(fb[b]["slope_priors"][:4]) + [sm["phi"][1], sm["phi"][0]])
        want = oracle.ssg_forecast(oracle.rng_philox(seed, c, 5), T, newX, beta[c], sig[c], fb,
                                   sg, np.zeros((len(blocks), 16)), st[-1])
        assert np.max(np.abs(fc[c] - want)) < 1e-8 * max(1.0, np.abs(want).max()), (key, c)
    eng.close()


def test_glob_lists_behind_the_look_ahead():
    """bsts's loop (ba_ss_draw_next + chain 0's draw and state path) on a list with both new
    models equals one round per call, every chain, bit for bit"""
    T, p, chains, seed = 90, 4, 6, 11
    X, y, _, obs = general_data(T, p, 2, [], seed=12, trig=[(12.0, [1.0])], intercept=3.0, missing_frac=0.02)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, [("intercept",), ("trig", 12.0, [1.0, 2.0]), ("ar", 1), ("semilocal",)])
    g0 = np.zeros(p, np.uint8)
    a = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    b = make_engine(chains, seed, y, X, obs, prior, blocks, sig_up, g0)
    a.ss_set_lookahead(8)
    for it in range(20):
        a.ss_draw_next()
        b.ss_sweep(1)
        for x, z in zip(a.get_states(), b.get_states()):
            assert np.array_equal(x, z), it
        assert np.array_equal(a.ss_get_state_draw(0), b.ss_get_state_draw(0)), it
        assert np.array_equal(a.ss_get_state_model(2, 1)["variances"], b.ss_get_state_model(2, 1)["variances"]), it
        u, v = a.ss_get_state_model(3, 3, suf=False), b.ss_get_state_model(3, 3, suf=False)
        assert np.array_equal(u["phi"], v["phi"]) and np.array_equal(u["variances"], v["variances"]), it
    a.close()
    b.close()
