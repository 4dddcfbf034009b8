"""Pins the CPU oracle (oracle/boom_oracle.c) against fixtures produced by the
compiled, unmodified reference (tests/golden/make_golden.py).  Runs anywhere:
no GPU, no /root/reference.

Tolerances: discrete outputs (inclusion indicators, permutations, integer
draws) bit-exact; scalar RNG transforms bit-exact (same libm); continuous
draws that go through Eigen-vectorised reductions in the reference <= 1e-9
relative.
"""
import os

import numpy as np
import pytest

from oracle_lib import ssvs_options

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-9


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def prior_of(g):
    return dict(b=g["prior_b"], ominv=g["prior_ominv"], df=float(g["prior_df"]),
                sigma_guess=float(g["prior_sigma_guess"]), pi=g["prior_pi"])


def opts_of(g):
    return ssvs_options(max_model_size=int(g["opt_max_model_size"]),
                        sigma_upper_limit=float(g["opt_sigma_upper_limit"]),
                        swap_threshold=float(g["opt_swap_threshold"]),
                        max_flips=int(g["opt_max_flips"]))


# --------------------------------------------------------------------- RNG
def test_rng_known_answers(oracle):
    g = load("kat_rng")
    seed = int(g["seed"])
    O = oracle
    assert np.array_equal(O.uniforms(O.rng_mt(seed), 512), g["uniform"])
    assert np.array_equal(O.seed_rngs(O.rng_mt(seed), 32), g["seed_rng"])
    assert np.array_equal(O.norms(O.rng_mt(seed), 2048), g["norm"])
    assert np.array_equal(O.exps(O.rng_mt(seed), 1024), g["exp"])
    for a, want in zip(g["gamma_shapes"], g["gamma"]):
        got = O.gammas(O.rng_mt(seed), float(a), float(g["gamma_rate"]), 1024)
        assert np.array_equal(got, want), a
    a, b, cut = g["trun_gamma_args"]
    assert np.array_equal(O.trun_gammas(O.rng_mt(seed), a, b, cut, 1024),
                          g["trun_gamma"])
    assert np.array_equal(O.random_ints(O.rng_mt(seed), 0, 511, 1024),
                          g["random_int"])
    assert np.array_equal(O.shuffles(O.rng_mt(seed), 512, 4), g["shuffle"])
    assert np.array_equal(O.rmultis(O.rng_mt(seed), g["rmulti_prob"], 512),
                          g["rmulti"])


def test_truncated_gamma_all_regimes(oracle):
    """rtrun_gamma_mt: rejection, adaptive rejection and slice regimes
    (distributions/trun_gamma.cpp:74-100), vectors of the compiled reference"""
    g = load("kat_trun_gamma")
    seed = int(g["seed"])
    for (a, b, cut), want in zip(g["cases"], g["draws"]):
        got = oracle.trun_gammas(oracle.rng_mt(seed), float(a), float(b), float(cut),
                                 want.shape[0])
        assert np.array_equal(got, want), (a, b, cut)
        assert np.all(want >= cut)
    for a, want in zip(g["small_shapes"], g["small_draws"]):   # shape < 0.3
        got = oracle.gammas(oracle.rng_mt(seed), float(a), float(g["small_rate"]),
                            want.shape[0])
        assert np.array_equal(got, want), a
    # a truncation point exactly at the mode is the reference's reported error
    import ctypes as C
    st = C.c_int(0)
    r = oracle.rng_mt(seed)
    oracle.lib.bo_rtrun_gamma(C.byref(r), 10.0, 2.0, 4.5, C.byref(st))
    assert st.value != 0


def test_philox_known_answers(oracle):
    """Published Philox4x32-10 test vectors (Random123 kat_vectors)."""
    import ctypes as C
    vec = [
        ((0, 0, 0, 0), (0, 0),
         (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
         (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in vec:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        oracle.lib.bo_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_philox_stream_layout(oracle):
    """uniform i of a stream = half (i&1) of block (i>>1), (x>>11)*2^-53;
    streams are random-access via pos."""
    a = oracle.uniforms(oracle.rng_philox(123, chain=5, stream=0), 64)
    b = oracle.uniforms(oracle.rng_philox(123, chain=5, stream=0, pos=17), 8)
    assert np.array_equal(a[17:25], b)
    assert np.all((a >= 0) & (a < 1))
    c = oracle.uniforms(oracle.rng_philox(123, chain=6, stream=0), 64)
    assert not np.any(a == c)


def test_state_stream_gives_every_normal_its_own_position(oracle):
    """Stream 2 (the normals of simulate_forward) in Philox mode: draw number s owns the
    positions [256 s, 256 (s + 1)), whatever the draws before it consumed, and draws 2 j and
    2 j + 1 are the Box-Muller pair of the two uniforms at position 512 j (round 6; one
    Kinderman-Ramage draw per slot before) -- the contract the device's stream_normals.h
    implements (the MT engine, pinned on the reference, reads in sequence through norm_rand; so
    does every other Philox stream)."""
    import ctypes as C
    from math import cos, log, sin, sqrt
    L = oracle.lib
    L.bo_rnorm.restype = C.c_double
    L.bo_rnorm.argtypes = [C.c_void_p, C.c_double, C.c_double]
    r = oracle.rng_philox(99, chain=3, stream=2)
    seq = [L.bo_rnorm(C.byref(r), 0.0, 1.0) for _ in range(200)]
    assert L.bo_rnorm(C.byref(r), 5.0, 0.0) == 5.0 and r.slot == 200      # sigma = 0: no draw, no slot
    assert r.pos == 200 * 256
    for i in (0, 1, 57, 198, 199):
        # the same draw from a stream positioned at its slot: a generator that STARTS on an odd
        # draw number still takes the second half of that number's pair
        one = oracle.rng_philox(99, chain=3, stream=2, pos=256 * i)
        assert L.bo_rnorm(C.byref(one), 0.0, 1.0) == seq[i]
        # ... which is the transform of the two uniforms at the pair's first slot (read through
        # a sequential stream id with the same key: stream 2 | the uniforms do not depend on the mode)
        u = oracle.uniforms(oracle.rng_philox(99, chain=3, stream=2, pos=256 * (i & ~1)), 2) \
            if hasattr(oracle, "uniforms") else None
        if u is not None:
            R, th = sqrt(-2.0 * log(1.0 - u[0])), 6.283185307179586 * u[1]
            assert abs(seq[i] - (R * sin(th) if i & 1 else R * cos(th))) < 1e-15 * max(1.0, abs(seq[i]))
    # another stream reads in sequence: the second normal starts where the first stopped
    s0 = oracle.rng_philox(99, chain=3, stream=0)
    a = L.bo_rnorm(C.byref(s0), 0.0, 1.0)
    used = s0.pos
    b = L.bo_rnorm(C.byref(s0), 0.0, 1.0)
    s1 = oracle.rng_philox(99, chain=3, stream=0, pos=used)
    assert L.bo_rnorm(C.byref(s1), 0.0, 1.0) == b and a != b and used in (2, 3, 4, 5, 6, 7, 8)


# ------------------------------------------------------------------ LinAlg
def test_linalg_known_answers(oracle):
    g = load("kat_linalg")
    oa = ol = ov = 0
    for i, n in enumerate(g["sizes"]):
        A = g["A"][oa:oa + n * n].reshape(n, n)
        Lw = g["L"][ol:ol + n * n].reshape(n, n)
        rhs = g["rhs"][ov:ov + n]
        x = g["x"][ov:ov + n]
        L, ok = oracle.chol(A)
        assert ok and relerr(L, Lw, 1e-6) < 1e-11
        ld, ok = oracle.logdet(A)
        assert ok and abs(ld - g["logdet"][i]) < 1e-11 * max(1, abs(ld))
        sol, ok = oracle.solve(A, rhs)
        assert ok and relerr(sol, g["sol"][ov:ov + n], 1e-6) < 1e-9
        assert abs(oracle.mdist(A, x) - g["mdist"][i]) < 1e-11 * g["mdist"][i]
        oa += n * n
        ol += n * n
        ov += n
    assert oracle.logdet(g["notpd"])[1] == bool(g["notpd_logdet_ok"]) is False
    assert oracle.solve(g["notpd"], np.ones(3))[1] == bool(g["notpd_solve_ok"])


def test_neregsuf(oracle):
    g = load("ssvs_c1")
    suf = oracle.neregsuf(g["X"], g["y"])
    n = g["X"].shape[0]
    assert relerr(suf["xtx"], g["xtx"], 1e-6) < 1e-12
    assert relerr(suf["xty"], g["xty"], 1e-6) < 1e-12
    assert abs(suf["yty"] - g["yty"]) < 1e-12 * g["yty"]
    assert abs(suf["sumy"] / n - g["ybar"]) < 1e-13
    assert relerr(suf["xsum"] / n, g["xbar"], 1e-6) < 1e-12


# -------------------------------------------------------------------- SSVS
def test_log_model_prob(oracle):
    g = load("kat_log_model_prob")
    n = float(g["n"])
    suf = dict(xtx=g["xtx"], xty=g["xty"], yty=float(g["yty"]), n=n,
               sumy=float(g["ybar"]) * n, xsum=g["xbar"] * n)
    prior = prior_of(g)
    got = oracle.log_model_prob(suf, prior, g["gammas"])
    assert relerr(got, g["logp"]) < 1e-12
    got3 = oracle.log_model_prob(suf, prior, g["gammas"], max_model_size=3)
    want3 = g["logp_max3"]
    assert np.array_equal(np.isneginf(got3), np.isneginf(want3))
    m = np.isfinite(want3)
    assert relerr(got3[m], want3[m]) < 1e-12


def test_logpri(oracle):
    """BregVsSampler::logpri() (the PosteriorSampler interface's log prior)"""
    g = load("kat_logpri")
    n = float(g["n"])
    suf = dict(xtx=g["xtx"], xty=g["xty"], yty=float(g["yty"]), n=n,
               sumy=float(g["ybar"]) * n, xsum=g["xbar"] * n)
    got = oracle.logpri(suf, prior_of(g), g["gammas"], g["betas"], g["sigsqs"])
    assert relerr(got, g["logpri"]) < 1e-12


@pytest.mark.parametrize("name", ["ssvs_c1", "ssvs_p64", "ssvs_collinear",
                                  "ssvs_general", "ssvs_maxflips", "ssvs_empty",
                                  "ssvs_tight_sigma", "ssvs_binding_sigma"])
def test_ssvs_sweeps_match_reference(oracle, name):
    g = load(name)
    suf = oracle.neregsuf(g["X"], g["y"])
    prior = prior_of(g)
    opts = opts_of(g)
    for i, seed in enumerate(g["seeds"]):
        o = oracle.ssvs_run(suf, prior, opts, ("mt", int(seed)),
                            g["init_gamma"], int(g["nsweeps"]),
                            want_margin=True)
        assert o["status"] == 0
        assert np.array_equal(o["gamma"], g["gamma"][i]), name
        assert relerr(o["beta"], g["beta"][i]) < RTOL
        assert relerr(o["sigsq"], g["sigsq"][i]) < RTOL
        # decisions were never within rounding of the accept boundary
        assert o["min_margin"] > 1e-7


def test_ssvs_reference_acceptance_criteria():
    """The reference's own statistical checks re-expressed on its draws
    (regression_spike_slab_test.cc:124-171, :207-257)."""
    g = load("ssvs_general")
    assert g["gamma"][0].sum(axis=1).max() <= 4          # max_model_size
    assert np.sqrt(g["sigsq"][0]).max() <= 1.08          # sigma upper limit
    g = load("ssvs_collinear")
    inc = g["gamma"][0][:, [1, 2, 3, 7]]
    assert inc.sum(axis=1).mean() > 0.9   # one of the collinear set is in
    g = load("ssvs_maxflips")
    ch = np.abs(np.diff(g["gamma"][0].astype(int), axis=0)).sum(axis=1)
    assert ch.max() <= 5 + 2              # <= max_flips (+ one swap move)


def test_convenience_ctor_priors(oracle):
    g = load("ssvs_ctors")
    suf = oracle.neregsuf(g["X"], g["y"])
    a = g["ctor1_args"]
    p1 = oracle.prior_ctor1(suf, a[0], a[1], a[2], bool(g["ctor1_flag"]))
    o = oracle.ssvs_run(suf, p1, ssvs_options(), ("mt", int(g["seed"])),
                        g["init_gamma"], int(g["nsweeps"]))
    assert np.array_equal(o["gamma"], g["gamma1"])
    assert relerr(o["beta"], g["beta1"]) < RTOL
    assert relerr(o["sigsq"], g["sigsq1"]) < RTOL
    a = g["ctor2_args"]
    p2 = oracle.prior_ctor2(suf, a[0], a[1], a[2], a[3], a[4],
                            bool(g["ctor2_flag"]))
    o = oracle.ssvs_run(suf, p2, ssvs_options(), ("mt", int(g["seed"])),
                        g["init_gamma"], int(g["nsweeps"]))
    assert np.array_equal(o["gamma"], g["gamma2"])
    assert relerr(o["beta"], g["beta2"]) < RTOL
    assert relerr(o["sigsq"], g["sigsq2"]) < RTOL


# ------------------------------------------------- SpikeSlabSampler (a11)
@pytest.mark.parametrize("name", ["sss_kind0_case0", "sss_kind0_case1",
                                  "sss_kind1_case0", "sss_kind1_case1"])
def test_spike_slab_sampler_matches_reference(oracle, name):
    g = load(name)
    n, p = g["X"].shape
    W = g["w"]
    xtx = (g["X"].T * W) @ g["X"]
    assert relerr(xtx, g["xtx"], 1e-6) < 1e-12      # WeightedRegSuf
    o = oracle.sss_run(g["xtx"], g["xty"], int(g["slab_kind"]), g["mu"], g["prec"],
                       g["pi"], ("mt", int(g["seed"])), g["init_gamma"], g["sigsq"],
                       max_model_size=int(g["max_model_size"]),
                       max_flips=int(g["max_flips"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < RTOL


# ------------------------------------------------------------- state space
@pytest.mark.parametrize("name", ["ss_t200", "ss_t200_missing", "ss_t3", "ss_t1"])
def test_state_space_sweeps_match_reference(oracle, name):
    g = load(name)
    ss = dict(zip([str(k) for k in g["ss_keys"]], [float(v) for v in g["ss_vals"]]))
    obs = g["observed"]
    o = oracle.ss_run(g["y"], g["X"], None if obs.all() else obs, prior_of(g),
                      opts_of(g), ss, ("mt", int(g["seed"])), g["init_gamma"],
                      int(g["nsweeps"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < RTOL
    assert relerr(o["sigsq"], g["sigsq"]) < RTOL
    assert relerr(o["level_sigsq"], g["level_sigsq"]) < RTOL
    assert np.max(np.abs(o["state"] - g["state"])) < 1e-9 * np.abs(g["state"]).max()


@pytest.mark.parametrize("name", ["ssm_level", "ssm_trend", "ssm_level_seasonal7",
                                  "ssm_trend_seasonal4_missing", "ssm_trend_seasonal12"])
def test_structural_sweeps_match_reference(oracle, name):
    """f2: regression + local level / local linear trend + seasonal state"""
    g = load(name)
    trend, ns = int(g["trend"]), int(g["nseasons"])
    spec = dict(trend=trend, nseasons=ns, var_df=g["var_df"],
                var_sigma_guess=g["var_sigma_guess"],
                var_sigma_upper_limit=g["var_sigma_upper_limit"],
                var_initial_sigma=g["var_initial_sigma"],
                initial_state_mean=g["initial_state_mean"],
                initial_state_variance=g["initial_state_variance"])
    obs = g["observed"]
    o = oracle.ssm_run(g["y"], g["X"], None if obs.all() else obs, prior_of(g), opts_of(g),
                       spec, ("mt", int(g["seed"])), g["init_gamma"], int(g["nsweeps"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < RTOL
    assert relerr(o["sigsq"], g["sigsq"]) < RTOL
    idx = [0] + ([1] if trend == 2 else []) + ([2] if ns > 0 else [])
    assert relerr(o["variances"][:, idx], g["variances"][:, idx], 1e-300) < RTOL
    assert np.max(np.abs(o["state"] - g["state"])) < 1e-9 * np.abs(g["state"]).max()


def ar_spec_of(g):
    return dict(trend=int(g["trend"]), nseasons=int(g["nseasons"]), var_df=g["var_df"],
                var_sigma_guess=g["var_sigma_guess"],
                var_sigma_upper_limit=g["var_sigma_upper_limit"],
                var_initial_sigma=g["var_initial_sigma"],
                initial_state_mean=g["initial_state_mean"],
                initial_state_variance=g["initial_state_variance"],
                ar=dict(lags=int(g["ar_lags"]), df=float(g["ar_df"]),
                        sigma_guess=float(g["ar_sigma_guess"]),
                        sigma_upper_limit=float(g["ar_sigma_upper_limit"]),
                        initial_sigma=float(g["ar_initial_sigma"]),
                        initial_phi=g["ar_initial_phi"]))


@pytest.mark.parametrize("name", ["ssm_level_ar1", "ssm_trend_seasonal4_ar2_missing",
                                  "ssm_level_ar3"])
def test_structural_ar_sweeps_match_reference(oracle, name):
    """f2: an ArStateModel block + ArPosteriorSampler after the trend / seasonal state.
    In the second and third case many of the accepted coefficient vectors have
    sum |phi| >= 1, where the reference decides stationarity by finding roots and the
    oracle by the step-down recursion."""
    g = load(name)
    spec = ar_spec_of(g)
    trend, ns = spec["trend"], spec["nseasons"]
    obs = g["observed"]
    o = oracle.ssm_run(g["y"], g["X"], None if obs.all() else obs, prior_of(g), opts_of(g),
                       spec, ("mt", int(g["seed"])), g["init_gamma"], int(g["nsweeps"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < RTOL
    assert relerr(o["sigsq"], g["sigsq"]) < RTOL
    idx = [0] + ([1] if trend == 2 else []) + ([2] if ns > 0 else [])
    assert relerr(o["variances"][:, idx], g["variances"][:, idx], 1e-300) < RTOL
    assert relerr(o["ar_phi"], g["ar_phi"]) < RTOL
    assert relerr(o["ar_sigsq"], g["ar_sigsq"], 1e-300) < RTOL
    assert np.max(np.abs(o["state"] - g["state"])) < 1e-9 * np.abs(g["state"]).max()


def test_ar_stationarity_known_answers(oracle):
    """ArModel::check_stationary on 400 coefficient vectors with sum |phi| around and
    above 1 (lags 1..8): the step-down recursion decides as the reference's root finder"""
    g = load("kat_ar_stationary")
    for phi, L, want in zip(g["phi"], g["lags"], g["stationary"]):
        v = np.ascontiguousarray(phi[:L])
        got = oracle.lib.bo_test_ar_check_stationary(int(L), v.ctypes.data_as(
            __import__("ctypes").POINTER(__import__("ctypes").c_double)))
        assert got == want, (phi[:L], want)


def test_two_sided_truncated_normal_known_answers(oracle):
    """rtrun_norm_2_mt (ArPosteriorSampler::draw_phi_univariate): both rejection
    samplers of the interior case and the Tn2Sampler tails, bit for bit"""
    import ctypes as C
    g = load("kat_trun_norm_2")
    oracle.lib.bo_rtrun_norm_2.restype = C.c_double
    oracle.lib.bo_rtrun_norm_2.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.POINTER(C.c_int)]
    for case, want in zip(g["cases"], g["draws"]):
        rng, st = oracle.rng_mt(int(g["seed"])), C.c_int(0)
        got = np.array([oracle.lib.bo_rtrun_norm_2(C.byref(rng), *[float(v) for v in case],
                                                   C.byref(st)) for _ in range(want.shape[0])])
        assert st.value == 0 and np.array_equal(got, want), case


def test_structural_forecast_known_answers(oracle):
    g = load("kat_structural_forecast")
    for trend, ns in g["shapes"]:
        key = "t%d_s%d" % (trend, ns)
        got = oracle.ssm_forecast(oracle.rng_mt(int(g["seed"])), g["newX"], g["beta"],
                                  float(g["sigsq_obs"]), int(trend), int(ns), g["sigsq"],
                                  g[key + "_final_state"])
        assert np.max(np.abs(got - g[key + "_forecast"])) < 1e-12, key


GENERAL_GOLDENS = ["ssg_seasonal_only", "ssg_ar_only", "ssg_weekly_annual",
                   "ssg_seasonal_first_missing", "ssg_duration_t0", "ssg_two_ar",
                   "ssg_level_and_trend", "ssg_big52"]


@pytest.mark.parametrize("name", GENERAL_GOLDENS)
def test_general_state_lists_match_reference(oracle, name):
    """f2, the general form: state models added in any order -- a seasonal-only and an
    autoregression-only model, seasonal blocks with season_duration > 1 (T / RQR switch at
    new_season) and a time_of_first_observation, two seasonal blocks (weekly + a 4-season
    cycle of duration 7), a seasonal block ahead of the level, two autoregression blocks,
    a local level beside a local linear trend, nseasons = 52 with duration 7 (m = 53)"""
    from cases import blocks_of
    g = load(name)
    blocks = blocks_of(g)
    obs = g["observed"]
    o = oracle.ssg_run(g["y"], g["X"], None if obs.all() else obs, prior_of(g), opts_of(g),
                       blocks, ("mt", int(g["seed"])), g["init_gamma"], int(g["nsweeps"]),
                       int(g["state_every"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < RTOL
    assert relerr(o["sigsq"], g["sigsq"]) < RTOL
    assert relerr(o["variances"], g["variances"], 1e-300) < RTOL
    assert relerr(o["phi"], g["phi"], 1e-300) < RTOL
    assert o["state"].shape == g["state"].shape
    assert np.max(np.abs(o["state"] - g["state"])) < 1e-9 * np.abs(g["state"]).max()


def test_general_forecast_known_answers(oracle):
    """simulate_forecast with seasonal blocks of duration > 1: the reference simulates
    forecast step i with the transition matrix and state errors of time T - 2 + i"""
    from cases import blocks_of
    g = load("kat_general_forecast")
    for key in g["shapes"]:
        key = str(key)
        blocks = blocks_of(g, key + "_")
        got = oracle.ssg_forecast(oracle.rng_mt(int(g["seed"])), int(g[key + "_T"]), g["newX"],
                                  g["beta"], float(g["sigsq_obs"]), blocks, g[key + "_sigsq"],
                                  g[key + "_phi"], g[key + "_final_state"])
        assert np.max(np.abs(got - g[key + "_forecast"])) < 1e-12, key


GLOB_GOLDENS = ["ssq_intercept_ar", "ssq_intercept_seasonal_missing", "ssq_trig_only", "ssq_trend_trig",
                "ssq_trig_level_seasonal", "ssq_two_trig_intercept", "ssq_semilocal",
                "ssq_seasonal_semilocal_missing", "ssq_semilocal_free_trig"]


@pytest.mark.parametrize("name", GLOB_GOLDENS)
def test_static_intercept_and_trig_state_models_match_reference(oracle, name):
    """f2 along its glob (VERDICT r5 task 8): StaticInterceptStateModel (one component, T = 1,
    no state error, no parameter: StaticInterceptStateModel.hpp:35-131) and TrigStateModel (a
    2 x 2 rotation per frequency, Z = 1 at every pair's first component, ONE variance for all
    components: TrigStateModel.cpp:130-223), alone, together, and in lists with the round-4
    models, missing observations included; SemilocalLinearTrendStateModel (level, slope, the
    slope's long-run mean; the level's variance sampler and the slope's NonzeroMeanAr1Sampler:
    SemilocalLinearTrend.cpp:29-272, NonzeroMeanAr1Sampler.cpp:51-155) with the AR(1) coefficient
    truncated to [-1, 1], to [0, 1] and not at all -- the compiled reference's draws (goldens of
    make_golden_structural_glob.py)."""
    from cases import blocks_of
    g = load(name)
    blocks = blocks_of(g)
    obs = g["observed"]
    o = oracle.ssg_run(g["y"], g["X"], None if obs.all() else obs, prior_of(g), opts_of(g),
                       blocks, ("mt", int(g["seed"])), g["init_gamma"], int(g["nsweeps"]),
                       int(g["state_every"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < RTOL
    assert relerr(o["sigsq"], g["sigsq"]) < RTOL
    assert relerr(o["variances"], g["variances"], 1e-300) < RTOL
    assert relerr(o["phi"], g["phi"], 1e-6) < RTOL       # (a semilocal trend's phi and mu)
    assert o["state"].shape == g["state"].shape
    assert np.max(np.abs(o["state"] - g["state"])) < 1e-9 * np.abs(g["state"]).max()


def test_glob_forecast_known_answers(oracle):
    """simulate_forecast with a static intercept (no error term, no draw) and trig blocks
    (2 nfreq error draws a step, rotations)"""
    from cases import blocks_of
    g = load("kat_glob_forecast")
    for key in g["shapes"]:
        key = str(key)
        blocks = blocks_of(g, key + "_")
        got = oracle.ssg_forecast(oracle.rng_mt(int(g["seed"])), int(g[key + "_T"]), g["newX"],
                                  g["beta"], float(g["sigsq_obs"]), blocks, g[key + "_sigsq"],
                                  g[key + "_phi"], g[key + "_final_state"])
        assert np.max(np.abs(got - g[key + "_forecast"])) < 1e-12, key


def test_general_form_repeats_the_template(oracle):
    """the block list [trend, seasonal(ns, 1), ar] is the template of rounds 2-3, draw for
    draw (both Philox and MT generators)"""
    from cases import bsts_priors, general_spec, structural_data, structural_spec
    X, y, _, obs = structural_data(90, 5, 2, 4, seed=8, missing_frac=0.05, ar_coef=[0.5])
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, 2, 4, ar_lags=1)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(5, np.uint8)
    blocks = general_spec(y, [("trend",), ("seasonal", 4, 1), ("ar", 1)])
    for setup in (("philox", 5, 3), ("mt", 77)):
        a = oracle.ssm_run(y, X, obs, prior, opts, spec, setup, g0, 15)
        b = oracle.ssg_run(y, X, obs, prior, opts, blocks, setup, g0, 15)
        assert a["status"] == 0 and b["status"] == 0
        assert np.array_equal(a["gamma"], b["gamma"])
        assert np.array_equal(a["beta"], b["beta"]) and np.array_equal(a["state"], b["state"])
        assert np.array_equal(a["variances"][:, 0], b["variances"][:, 0, 0])
        assert np.array_equal(a["variances"][:, 1], b["variances"][:, 0, 1])
        assert np.array_equal(a["variances"][:, 2], b["variances"][:, 1, 0])
        assert np.array_equal(a["ar_phi"][:, 0], b["phi"][:, 2, 0])
        assert np.array_equal(a["ar_sigsq"], b["variances"][:, 2, 0])


# ------------------------------------------------------------------ probit
def test_truncated_normal_known_answers(oracle):
    """rtrun_norm_mt: rejection from the normal (cut below the mean) and the
    bounded adaptive rejection sampler TnSampler (distributions/trun_norm.cpp)"""
    g = load("kat_trun_norm")
    for (mu, sg, cut, ab), want in zip(g["cases"], g["draws"]):
        got = oracle.trun_norms(oracle.rng_mt(int(g["seed"])), float(mu), float(sg), float(cut),
                                int(ab), want.shape[0])
        assert np.array_equal(got, want), (mu, sg, cut, ab)


@pytest.mark.parametrize("name", ["probit_bernoulli", "probit_binomial8_clt3",
                                  "probit_binomial12"])
def test_probit_spike_slab_matches_reference(oracle, name):
    """f3 (probit): BinomialProbitSpikeSlabSampler.  The latent data carry rounding
    differences from sweep to sweep and amplify them; 12 sweeps stay below 1e-9."""
    g = load(name)
    p = g["X"].shape[1]
    o = oracle.probit_run(g["X"], g["y"], g["ntrials"], dict(mu=g["mu"], prec=g["prec"]),
                          g["pi"], ("mt", int(g["seed"])), g["init_gamma"], np.zeros(p),
                          int(g["nsweeps"]), clt_threshold=int(g["clt_threshold"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < 1e-9


@pytest.mark.parametrize("name", ["logit_bernoulli", "logit_binomial4",
                                  "logit_bernoulli_p24_maxflips", "logit_binomial60_large_sample",
                                  "logit_binomial200_large_sample"])
def test_logit_spike_slab_matches_reference(oracle, name):
    """f3 (logit): BinomialLogitSpikeSlabSampler with its auxiliary-mixture imputer"""
    g = load(name)
    p = g["X"].shape[1]
    o = oracle.logit_run(g["X"], g["y"], g["ntrials"], dict(mu=g["mu"], prec=g["prec"]),
                         g["pi"], ("mt", int(g["seed"])), g["init_gamma"], np.zeros(p),
                         int(g["nsweeps"]), clt_threshold=int(g["clt_threshold"]),
                         max_flips=int(g["max_flips"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < 1e-9


def _golden_mix(g):
    return dict(counts=g["mix_counts"], ncomp=g["mix_ncomp"], mu=g["mix_mu"], sigma=g["mix_sigma"],
                weight=g["mix_weight"], largest_index=int(g["mix_largest_index"]))


@pytest.mark.parametrize("name", ["poisson_small_counts", "poisson_exposure", "poisson_large_counts",
                                  "poisson_p24_maxflips"])
def test_poisson_spike_slab_matches_reference(oracle, name):
    """f3 (Poisson): PoissonRegressionSpikeSlabSampler -- Cheng's beta draws for the last
    event time, the exponential / extreme-value draw past the interval, the unmixing of
    both NegLogGamma residuals against the reference table's mixtures (fixture data),
    SpikeSlabSampler on the complete-data sufficient statistics"""
    g = load(name)
    p = g["X"].shape[1]
    o = oracle.poisson_run(g["X"], g["y"], g["exposure"], dict(mu=g["mu"], prec=g["prec"]), g["pi"],
                           _golden_mix(g), ("mt", int(g["seed"])), g["init_gamma"], np.zeros(p),
                           int(g["nsweeps"]), max_flips=int(g["max_flips"]))
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], g["gamma"])
    assert relerr(o["beta"], g["beta"]) < 1e-9


def test_impute_state_known_answer(oracle):
    g = load("kat_impute_state")
    o = oracle.ss_impute_state(g["y"], g["X"], g["observed"], g["beta"],
                               g["gamma"], float(g["sigsq_obs"]),
                               float(g["sigsq_level"]), float(g["a0"]),
                               float(g["P0"]), oracle.rng_mt(int(g["seed"])))
    assert np.max(np.abs(o["state"] - g["state"])) < 1e-11
    assert o["level_n"] == float(g["level_n"])
    assert abs(o["level_sumsq"] - float(g["level_sumsq"])) < 1e-11 * float(g["level_sumsq"])


@pytest.mark.parametrize("name", ["adaptive_c1", "adaptive_p150", "adaptive_collinear",
                                  "adaptive_options", "adaptive_wide_start80",
                                  "adaptive_wide_growth"])
def test_adaptive_sampler_matches_reference(oracle, name):
    """AdaptiveSpikeSlabRegressionSampler (what lm.spike runs for p > 100): the
    oracle's restatement on the reference's engine and seed against the
    reference's own draws: inclusion indicators identical, beta / sigma^2 to
    rounding."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    suf = dict(xtx=g["xtx"], xty=g["xty"], yty=float(g["yty"]), n=float(g["n"]),
               sumy=float(g["sumy"]), xsum=g["xsum"])
    prior = dict(b=g["prior_b"], ominv=g["prior_ominv"], df=float(g["prior_df"]),
                 sigma_guess=float(g["prior_sigma_guess"]), pi=g["prior_pi"])
    opts = ssvs_options(max_model_size=int(g["opt_max_model_size"]),
                        sigma_upper_limit=float(g["opt_sigma_upper_limit"]))
    p = len(suf["xty"])
    want = np.unpackbits(g["gamma"], axis=1)[:, :p]
    nsw = want.shape[0]
    o = oracle.adaptive_run(suf, prior, opts, ("mt", int(g["seed"])), g["init_gamma"], nsw,
                            int(g["max_flips"]), float(g["step_size"]), float(g["target"]),
                            want_margin=True)
    assert o["status"] == 0
    assert np.array_equal(o["gamma"], want)
    assert np.max(np.abs(o["beta"] - g["beta"]) / np.maximum(np.abs(g["beta"]), 1e-3)) < 1e-11
    assert np.max(np.abs(o["sigsq"] - g["sigsq"]) / g["sigsq"]) < 1e-11
    assert o["min_margin"] > 1e-9 and o["min_multi_margin"] > 1e-12


def test_simulate_forecast_known_answer(oracle):
    """StateSpaceRegressionModel::simulate_forecast (local level + regression)
    against the reference's own output for fixed parameters and final state."""
    g = np.load(os.path.join(GOLD, "kat_forecast.npz"))
    got = oracle.ss_forecast(oracle.rng_mt(int(g["seed"])), g["newX"], g["beta"],
                             float(g["sigsq_obs"]), float(g["sigsq_level"]),
                             float(g["final_state"]))
    assert np.max(np.abs(got - g["forecast"])) < 1e-13
