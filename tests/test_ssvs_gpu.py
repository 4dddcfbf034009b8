"""GPU parity: the HIP SSVS path (through the C-ABI) against the CPU oracle on
the same seeded inputs and the same Philox streams.

Bars: inclusion indicators bit-exact; beta, sigma^2 within 1e-8 relative (the
kernel keeps updated Cholesky factors and evaluates proposals in O(k^2); the
oracle refactors from scratch like the reference -- the two differ by
rounding only).  Stated fp64 tolerance for continuous draws: RTOL below.
"""
import os

import numpy as np
import pytest

from cases import regression_data, spike_slab_prior, suf_from_xy
from oracle_lib import ssvs_options

pytestmark = pytest.mark.gpu
RTOL = 1e-8
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def make_engine(chains, seed, suf=None, X=None, y=None, prior=None, opts=None,
                g0=None, tuning=None, **kw):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed, **kw)
    if tuning:
        eng.set_tuning(**tuning)
    if X is not None:
        eng.build_suf_from_xy(X, y)
    else:
        eng.upload_suf(suf["xtx"], suf["xty"], suf["yty"], suf["n"],
                       suf["sumy"] / suf["n"], suf["xsum"] / suf["n"])
    opts = opts or ssvs_options()
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], max_model_size=opts["max_model_size"],
                   sigma_upper_limit=opts["sigma_upper_limit"])
    eng.set_options(max_flips=opts["max_flips"],
                    swap_threshold=opts["swap_threshold"])
    eng.set_state(g0)
    return eng


def compare_chain_by_chain(oracle, eng, suf, prior, opts, seed, g0, nsweeps,
                           chains_to_check, step=1, chain_offset=0):
    """sweep `step` at a time and compare every recorded draw"""
    ora = {c: oracle.ssvs_run(suf, prior, opts, ("philox", seed, chain_offset + c),
                              g0, nsweeps, want_margin=True)
           for c in chains_to_check}
    for c in chains_to_check:
        assert ora[c]["status"] == 0
    done = 0
    while done < nsweeps:
        eng.sweep(step)
        done += step
        gam, beta, sig = eng.get_states()
        for c in chains_to_check:
            o = ora[c]
            assert np.array_equal(gam[c], o["gamma"][done - 1]), (c, done)
            assert relerr(beta[c], o["beta"][done - 1]) < RTOL, (c, done)
            assert abs(sig[c] - o["sigsq"][done - 1]) < RTOL * sig[c], (c, done)
    return ora


def test_suf_kernel_matches_reference_xtx():
    """a1: the MFMA X'X build against the reference's NeRegSuf (golden)."""
    import boom_amd
    g = np.load(os.path.join(GOLD, "ssvs_c1.npz"))
    eng = boom_amd.Engine(1)
    eng.build_suf_from_xy(g["X"], g["y"])
    s = eng.get_suf()
    assert relerr(s["xtx"], g["xtx"], 1e-6) < 1e-12
    assert relerr(s["xty"], g["xty"], 1e-6) < 1e-12
    assert abs(s["yty"] - g["yty"]) < 1e-12 * g["yty"]
    assert abs(s["ybar"] - g["ybar"]) < 1e-12
    assert relerr(s["xbar"], g["xbar"], 1e-6) < 1e-12
    # ragged sizes (n, p not multiples of the tile), shapes changing on one engine
    for n, p, seed in [(333, 71, 21), (64, 5, 2), (500, 130, 3), (333, 71, 21)]:
        X, y, _ = regression_data(n, p, min(5, p - 1), seed=seed)
        eng.build_suf_from_xy(X, y)
        s = eng.get_suf()
        ref = X.T @ X
        assert np.max(np.abs(s["xtx"] - ref)) < 1e-12 * np.abs(ref).max(), (n, p)
        assert np.array_equal(s["xtx"], s["xtx"].T)
    X, y, _ = regression_data(333, 71, 5, seed=21)
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    assert relerr(s["xtx"], X.T @ X, 1e-6) < 1e-12
    assert relerr(s["xty"], X.T @ y, 1e-6) < 1e-12


def test_log_model_prob_matches_reference_golden():
    """a5/a6 on the device against the reference's own numbers."""
    import boom_amd
    g = np.load(os.path.join(GOLD, "kat_log_model_prob.npz"))
    eng = boom_amd.Engine(1)
    eng.upload_suf(g["xtx"], g["xty"], float(g["yty"]), float(g["n"]),
                   float(g["ybar"]), g["xbar"])
    eng.set_priors(g["prior_b"], g["prior_ominv"], g["prior_pi"],
                   float(g["prior_df"]), float(g["prior_sigma_guess"]))
    got = eng.log_model_prob(g["gammas"])
    assert relerr(got, g["logp"]) < 1e-11
    eng.set_priors(g["prior_b"], g["prior_ominv"], g["prior_pi"],
                   float(g["prior_df"]), float(g["prior_sigma_guess"]),
                   max_model_size=3)
    got3 = eng.log_model_prob(g["gammas"])
    want3 = g["logp_max3"]
    assert np.array_equal(np.isneginf(got3), np.isneginf(want3))
    m = np.isfinite(want3)
    assert relerr(got3[m], want3[m]) < 1e-11


def test_c1_every_sweep(oracle):
    """C1 sizes (n=1000, p=20), draws compared after every single sweep:
    the reference's own calling pattern (one draw() per call)."""
    X, y, _ = regression_data(1000, 20, 6, seed=1)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 5)
    g0 = np.zeros(20, np.uint8)
    g0[0] = 1
    seed = 8675309
    eng = make_engine(8, seed, X=X, y=y, prior=prior, g0=g0)
    compare_chain_by_chain(oracle, eng, suf, prior, ssvs_options(), seed, g0, 60,
                           range(8), step=1)


def test_p64_batched_sweeps(oracle):
    X, y, _ = regression_data(600, 64, 12, seed=2)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 12)
    g0 = np.zeros(64, np.uint8)
    g0[0] = 1
    seed = 7
    eng = make_engine(32, seed, suf=suf, prior=prior, g0=g0)
    ora = compare_chain_by_chain(oracle, eng, suf, prior, ssvs_options(), seed, g0,
                                 100, [0, 5, 31], step=25)
    s = eng.get_summaries()
    assert s["sweeps"] == 32 * 100
    assert s["min_margin"] > 1e-9
    assert min(o["min_margin"] for o in ora.values()) > 1e-9


def test_collinear_swap_move(oracle):
    """correlated columns: the correlation swap move with real candidates
    (BregVsSampler::attempt_swap; regression_spike_slab_test.cc:207-257)."""
    X, y, _ = regression_data(500, 30, 2, seed=3, collinear=[1, 2, 3, 7])
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 3)
    g0 = np.zeros(30, np.uint8)
    g0[0] = 1
    seed = 11
    eng = make_engine(16, seed, suf=suf, prior=prior, g0=g0)
    compare_chain_by_chain(oracle, eng, suf, prior, ssvs_options(), seed, g0, 200,
                           range(16), step=50)
    # the swap move mattered: the same seeds without it give other chains
    eng2 = make_engine(16, seed, suf=suf, prior=prior,
                       opts=ssvs_options(swap_threshold=1.0), g0=g0)
    eng2.sweep(200)
    assert not np.array_equal(eng.get_states()[0], eng2.get_states()[0])


def test_general_priors_and_limits(oracle):
    """non-zero prior mean on every coefficient (exact path), max_model_size,
    truncated sigma draw, low swap threshold, no forced intercept, several
    variables included at the start."""
    X, y, _ = regression_data(40, 8, 3, seed=4)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 3, prior_mean=np.linspace(-1, 1, 8),
                             force_intercept=False)
    opts = ssvs_options(max_model_size=4, sigma_upper_limit=1.08,
                        swap_threshold=0.1)
    g0 = np.zeros(8, np.uint8)
    g0[:3] = 1
    seed = 5
    eng = make_engine(8, seed, suf=suf, prior=prior, opts=opts, g0=g0)
    compare_chain_by_chain(oracle, eng, suf, prior, opts, seed, g0, 200, range(8),
                           step=40)
    gam, _, sig = eng.get_states()
    assert gam.sum(axis=1).max() <= 4
    assert np.sqrt(sig).max() <= 1.08


def test_max_flips_and_suppressed_selection(oracle):
    X, y, _ = regression_data(40, 8, 3, seed=4)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 3, force_intercept=False,
                             prior_mean=np.zeros(8))
    g0 = np.zeros(8, np.uint8)
    g0[:3] = 1
    for mf in (5, 0):
        opts = ssvs_options(max_flips=mf)
        eng = make_engine(4, 9, suf=suf, prior=prior, opts=opts, g0=g0)
        compare_chain_by_chain(oracle, eng, suf, prior, opts, 9, g0, 50, range(4),
                               step=10)


def test_empty_and_full_models(oracle):
    """edge cases: the empty model (k = 0 closed form) reachable and a start
    that needs make_valid (pi_0 = 1 but intercept excluded)."""
    X, y, _ = regression_data(200, 6, 0, seed=8, intercept=False, noise_sd=1.0)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 1, force_intercept=False, prior_mean=np.zeros(6))
    g0 = np.zeros(6, np.uint8)
    eng = make_engine(8, 3, suf=suf, prior=prior, g0=g0)
    ora = compare_chain_by_chain(oracle, eng, suf, prior, ssvs_options(), 3, g0, 80,
                                 range(8), step=20)
    assert min(o["gamma"].sum(axis=1).min() for o in ora.values()) == 0
    # illegal start legalised by make_valid
    X, y, _ = regression_data(200, 6, 2, seed=9)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 2)  # pi_0 = 1
    g0 = np.zeros(6, np.uint8)        # intercept excluded: log prior = -inf
    eng = make_engine(4, 3, suf=suf, prior=prior, g0=g0)
    compare_chain_by_chain(oracle, eng, suf, prior, ssvs_options(), 3, g0, 20,
                           range(4), step=5)


def test_capacity_escalation(oracle):
    """a dense posterior (40 true signals, p = 64) started from the intercept
    alone: chains outgrow the 32-variable working set mid-run, are stopped at a
    sweep boundary and resumed with larger capacities -- invisibly to the
    draws, which still match the oracle sweep for sweep."""
    import boom_amd
    X, y, _ = regression_data(800, 64, 40, seed=13)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 40)
    g0 = np.zeros(64, np.uint8)
    g0[0] = 1
    eng = make_engine(8, 21, suf=suf, prior=prior, g0=g0)
    ora = compare_chain_by_chain(oracle, eng, suf, prior, ssvs_options(), 21, g0, 60,
                                 range(8), step=12)
    assert max(o["gamma"].sum(axis=1).max() for o in ora.values()) > 34
    # asynchronous launches queued behind a chain that stopped: nothing is lost
    eng2 = make_engine(8, 21, suf=suf, prior=prior, g0=g0)
    for _ in range(5):
        eng2.sweep(12, sync=False)
    eng2.sync()
    a, b = eng.get_states(), eng2.get_states()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert eng2.get_summaries()["sweeps"] == 8 * 60
    # a pinned capacity turns the same situation into a reported error
    eng3 = make_engine(8, 21, suf=suf, prior=prior, g0=g0, max_model_size_hint=16)
    with pytest.raises(boom_amd.BoomAmdError) as ei:
        eng3.sweep(60)
    assert "working capacity" in str(ei.value)
    # ... and it stays reported: an accessor after the failed check still begins with a
    # full check (only a CLEAN ba_sync lets the next one be a bare stream wait)
    for _ in range(2):
        with pytest.raises(boom_amd.BoomAmdError) as ei:
            eng3.get_state(0)
        assert "working capacity" in str(ei.value)
    with pytest.raises(boom_amd.BoomAmdError):
        eng3.sync()
    # a clean engine: accessors in a row, then work, then accessors again see the new draw
    g1, b1, s1 = eng.get_state(3)
    g2, b2, s2 = eng.get_state(3)
    assert np.array_equal(g1, g2) and np.array_equal(b1, b2) and s1 == s2
    eng.sweep(1, sync=False)
    g3, b3, s3 = eng.get_state(3)
    assert s3 != s1
    assert np.array_equal(eng.get_states()[1][3], b3)


def test_chain_offset_sharding(oracle):
    """chains keyed by GLOBAL id: an engine holding chains [8, 12) draws what
    chains 8..11 of a single big engine draw (multi-GPU sharding contract)."""
    X, y, _ = regression_data(300, 16, 4, seed=6)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 4)
    g0 = np.zeros(16, np.uint8)
    g0[0] = 1
    big = make_engine(12, 77, suf=suf, prior=prior, g0=g0)
    shard = make_engine(4, 77, suf=suf, prior=prior, g0=g0, chain_offset=8)
    big.sweep(30)
    shard.sweep(30)
    gb, bb, sb = big.get_states()
    gs, bs, ss = shard.get_states()
    assert np.array_equal(gb[8:], gs)
    assert np.array_equal(bb[8:], bs)
    assert np.array_equal(sb[8:], ss)


def test_error_reporting_matches_reference_messages():
    """bad call sequences and arguments are rejected with messages (a sigma
    upper limit tighter than the mode, an error in round 1, is now served:
    test_truncated_gamma_gpu.py)."""
    import boom_amd
    X, y, _ = regression_data(100, 5, 2, seed=10)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, 2)
    g0 = np.zeros(5, np.uint8)
    g0[0] = 1
    eng2 = boom_amd.Engine(2)
    with pytest.raises(boom_amd.BoomAmdError):
        eng2.sweep(1)  # no data, no priors
    # draw records: only after ba_enable_draws, only for chains the engine owns,
    # and a sweep() longer than the record is refused
    eng3 = make_engine(2, 1, suf=suf, prior=prior, g0=g0, chain_offset=10)
    eng3.sweep(3)
    with pytest.raises(boom_amd.BoomAmdError):
        eng3.get_draws(0, 3)
    eng3.enable_draws(4)
    eng3.sweep(4)
    gam, beta, sig = eng3.get_draws(1, 4)   # local index, like get_state
    assert gam.shape == (4, 5) and np.all(sig > 0) and np.all(beta[gam == 0] == 0)
    g1, b1, s1 = eng3.get_state(1)
    assert np.array_equal(gam[3], g1) and np.array_equal(beta[3], b1) and sig[3] == s1
    with pytest.raises(boom_amd.BoomAmdError):
        eng3.get_draws(2, 4)       # index outside [0, 2)
    with pytest.raises(boom_amd.BoomAmdError):
        eng3.sweep(5)              # more sweeps than record slots


def test_rank_deficient_model_is_the_reference_error(oracle):
    """a model whose posterior precision is singular (a zero column, no prior
    precision).  With model selection on, the reference stops in
    draw_model_indicators ("did not start with a legal configuration",
    BregVsSampler.cpp:364-370); with it off, draw_beta's factorisation fails, the
    reference re-enters draw() up to 10 times -- each attempt fails the same
    way on the same matrix -- and reports "not positive definite"
    (BregVsSampler.cpp:339-350).  The engine reports the same two errors."""
    import boom_amd
    X, y, _ = regression_data(100, 6, 2, seed=10)
    X[:, 3] = 0.0
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 2)
    prior["ominv"] = np.zeros((6, 6))
    g0 = np.ones(6, np.uint8)
    for max_flips, want, msg in ((0, 1, "not positive definite"),
                                 (-1, 3, "did not start with a legal configuration")):
        opts = ssvs_options(max_flips=max_flips)
        o = oracle.ssvs_run(suf, prior, opts, ("philox", 3, 0), g0, 3)
        assert o["status"] == want
        eng = make_engine(3, 3, suf=suf, prior=prior, opts=opts, g0=g0)
        with pytest.raises(boom_amd.BoomAmdError) as ei:
            eng.sweep(3)
        assert msg in str(ei.value)


def test_logpri_matches_oracle(oracle):
    """PosteriorSampler::logpri() of the chains' current states"""
    X, y, _ = regression_data(300, 16, 4, seed=6)
    suf = oracle.neregsuf(X, y)
    pm = np.zeros(16)
    pm[[0, 2, 9]] = [0.4, -0.3, 0.2]
    prior = spike_slab_prior(suf, 4, prior_mean=pm)
    g0 = np.zeros(16, np.uint8)
    g0[0] = 1
    eng = make_engine(5, 7, suf=suf, prior=prior, g0=g0)
    eng.sweep(25)
    gam, beta, sig = eng.get_states()
    want = oracle.logpri(suf, prior, gam, beta, sig)
    got = np.array([eng.logpri(c) for c in range(5)])
    assert relerr(got, want) < 1e-12


def test_summaries_and_traces(oracle):
    X, y, _ = regression_data(300, 16, 4, seed=6)
    suf = oracle.neregsuf(X, y)
    prior = spike_slab_prior(suf, 4)
    g0 = np.zeros(16, np.uint8)
    g0[0] = 1
    chains, nsw = 6, 40
    eng = make_engine(chains, 5, suf=suf, prior=prior, g0=g0)
    eng.enable_traces(nsw)
    eng.sweep(nsw)
    tr = eng.get_traces(nsw)
    s = eng.get_summaries()
    inc = np.zeros(16)
    bsum = np.zeros(16)
    for c in range(chains):
        o = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", 5, c), g0, nsw)
        inc += o["gamma"].sum(axis=0)
        bsum += o["beta"].sum(axis=0)
        assert relerr(tr["sigsq"][c], o["sigsq"]) < RTOL
        assert np.array_equal(tr["model_size"][c], o["gamma"].sum(axis=1))
    assert np.array_equal(s["inclusion_count"], inc)
    assert relerr(s["beta_sum"], bsum) < 1e-7
    assert s["sweeps"] == chains * nsw


def test_full_size_properties():
    """BASELINE config 2 shape (p=512, 1024 chains) through size-independent
    properties: valid states, forced intercept always in, signals found,
    inclusion frequencies of nulls small, sigma^2 near the truth."""
    n, p, nsig, chains = 10000, 512, 16, 1024
    X, y, _ = regression_data(n, p, nsig, seed=8675309)
    import boom_amd
    eng = boom_amd.Engine(chains, seed=1)
    eng.build_suf_from_xy(X, y)
    suf = eng.get_suf()
    suf2 = dict(xtx=suf["xtx"], xty=suf["xty"], yty=suf["yty"], n=suf["n"],
                sumy=suf["ybar"] * suf["n"], xsum=suf["xbar"] * suf["n"])
    prior = spike_slab_prior(suf2, 16)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.sweep(30)
    eng.reset_summaries()
    eng.sweep(30)
    gam, beta, sig = eng.get_states()
    assert gam[:, 0].all()
    assert gam[:, :nsig].mean() > 0.99
    assert gam[:, nsig:].mean() < 0.01
    assert np.all(beta[gam == 0] == 0.0)
    assert abs(np.sqrt(sig).mean() - 1.0) < 0.05
    s = eng.get_summaries()
    assert s["sweeps"] == chains * 30
    assert abs(s["k_sum"] / s["sweeps"] - gam.sum(axis=1).mean()) < 1.0
