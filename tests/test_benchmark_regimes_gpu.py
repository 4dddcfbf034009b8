"""Oracle parity in the regimes bench.py and the config tools actually run
("the benchmarked regime is not the regime the oracle checks"):

* C2  n=1e4, p=512, an engine of 1024 chains, default walk policy (proposal
      tables + forked quiet sweeps + two slots, 2 wavefronts per chain), burn-in
      and then ONE long launch with every draw recorded; chains {0, 1, 511, 1023}
      compared with the oracle draw by draw.
* C3  T=2000, p=100, 1024 chains (normals spanning many stream windows), and
      ragged / long T (second y* panel, T not a multiple of 64).
* C4  one shard at p=4096 (properties + oracle-compared chains).
* a0  the convenience constructors' priors as assembled by the engine.

Bars as everywhere: gamma bit-exact, continuous draws within 1e-8 relative.
All through the C-ABI (boom_amd.capi).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from cases import bsts_priors, regression_data, spike_slab_prior, state_space_data
from oracle_lib import ssvs_options

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def _engine_suf(eng):
    s = eng.get_suf()
    return dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"],
                sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])


def _check_draws(tag, draws, o, first, count):
    gam, beta, sig = draws
    for s in range(count):
        assert np.array_equal(gam[s], o["gamma"][first + s]), (tag, s)
        assert relerr(beta[s], o["beta"][first + s]) < RTOL, (tag, s)
        assert abs(sig[s] - o["sigsq"][first + s]) < RTOL * sig[s], (tag, s)


def test_c2_benchmark_regime_draw_by_draw(oracle):
    """BASELINE configs[1] exactly as bench.py runs it, 500-sweep launch."""
    import boom_amd
    n, p, nsig, chains, seed = 10000, 512, 16, 1024, 8675309
    burn, nsw = 200, 500
    X, y, _ = regression_data(n, p, nsig, seed=8675309)
    eng = boom_amd.Engine(chains, seed=seed)
    eng.build_suf_from_xy(X, y)
    suf = _engine_suf(eng)
    prior = spike_slab_prior(suf, nsig)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.sweep(burn)
    eng.reset_summaries()
    eng.enable_draws(nsw)
    eng.sweep(nsw)          # ONE launch: tables, forks, slots, kept model blocks
    check = [0, 1, 511, 1023]
    draws = {c: eng.get_draws(c, nsw) for c in check}
    sm = eng.get_summaries()

    def run(c):
        return oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", seed, c), g0,
                               burn + nsw, want_margin=True)
    with ThreadPoolExecutor(4) as ex:
        ora = dict(zip(check, ex.map(run, check)))
    for c in check:
        assert ora[c]["status"] == 0
        _check_draws(("c2", c), draws[c], ora[c], burn, nsw)
    # the decisions were safe: the smallest |log u - delta| seen by any of the
    # 1024 chains over the launch is far above the ~1e-12 rounding difference
    assert sm["sweeps"] == chains * nsw
    assert sm["min_margin"] > 1e-9
    assert min(o["min_margin"] for o in ora.values()) > 1e-9
    assert 15.5 < sm["k_sum"] / sm["sweeps"] < 17.5
    assert sm["slot_hits"] > 0          # the two-slot scheme was in play
    # the end state is the last recorded draw
    gam, beta, sig = eng.get_states()
    for c in check:
        assert np.array_equal(gam[c], draws[c][0][-1])
        assert np.array_equal(beta[c], draws[c][1][-1])


def _ss_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"],
                           ss["level_sigma_upper_limit"], ss["initial_state_mean"],
                           ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(g0)
    return eng


def _ss_compare(oracle, T, p, nsig, chains, check, nsw, seed, data_seed, missing=0.0):
    X, y, _, obs = state_space_data(T, p, nsig, seed=data_seed, missing_frac=missing)
    prior, ss, sig_up = bsts_priors(X, y, max(nsig, 1))
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = _ss_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0)

    def run(c):
        return oracle.ss_run(y, X, obs, prior, opts, ss, ("philox", seed, c), g0, nsw)
    with ThreadPoolExecutor(len(check)) as ex:
        ora = dict(zip(check, ex.map(run, check)))
    for s in range(nsw):
        eng.ss_sweep(1)
        gam, beta, sig = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            tag = (T, p, c, s)
            assert np.array_equal(gam[c], o["gamma"][s]), tag
            assert relerr(beta[c], o["beta"][s]) < RTOL, tag
            assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], tag
            st = eng.ss_get_state(c)
            assert abs(st["level_sigsq"] - o["level_sigsq"][s]) < RTOL * st["level_sigsq"], tag
            assert np.max(np.abs(st["state"] - o["state"][s])) < 1e-8 * np.abs(o["state"][s]).max(), tag


def test_c3_benchmark_shape_every_sweep(oracle):
    """BASELINE configs[2]: T=2000, p=100, 1024 chains; ~4000 normals per sweep
    span ~18 stream windows / 9 rounds of the two-wave hand-off."""
    _ss_compare(oracle, T=2000, p=100, nsig=5, chains=1024, check=[0, 1023], nsw=5,
                seed=4, data_seed=8675309)


@pytest.mark.parametrize("T,missing", [(2500, 0.0), (2049, 0.03), (4100, 0.0), (2048, 0.02), (2033, 0.0),
                                        (65, 0.0), (63, 0.1), (20, 0.0), (2, 0.0)])
def test_state_space_long_and_ragged_T(oracle, T, missing):
    """T beyond the lane-major kernel's 2048 steps (the natural-layout kernel), exactly 2048
    and a ragged last thread, T = 64 k + 1, T < 64, T = 2 (the level variance comes from the
    slice sampler)."""
    _ss_compare(oracle, T=T, p=8, nsig=3, chains=5, check=[0, 4], nsw=4, seed=17,
                data_seed=100 + T, missing=missing)


def test_c4_shard_p4096(oracle):
    """One shard of BASELINE configs[3] (p=4096, 32 signals): XtX = 134 MB lives
    in HBM / Infinity Cache, permutation and gamma arrays are 4096 long.  Size-
    independent properties for the shard, oracle parity for two chains."""
    import boom_amd
    n, p, nsig, chains, seed, nsw = 8192, 4096, 32, 128, 3, 24
    X, y, _ = regression_data(n, p, nsig, seed=8675309)
    eng = boom_amd.Engine(chains, seed=seed)
    eng.build_suf_from_xy(X, y)
    suf = _engine_suf(eng)
    ref = X.T @ X
    assert np.max(np.abs(suf["xtx"] - ref)) < 1e-11 * np.abs(ref).max()
    del ref
    prior = spike_slab_prior(suf, nsig)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.enable_draws(nsw)
    eng.sweep(nsw)
    check = [0, chains - 1]
    draws = {c: eng.get_draws(c, nsw) for c in check}

    def run(c):
        return oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", seed, c), g0,
                               nsw, want_margin=True)
    with ThreadPoolExecutor(2) as ex:
        ora = dict(zip(check, ex.map(run, check)))
    for c in check:
        assert ora[c]["status"] == 0
        _check_draws(("c4", c), draws[c], ora[c], 0, nsw)
    gam, beta, sig = eng.get_states()
    assert gam[:, 0].all()
    assert gam[:, :nsig].mean() > 0.99
    assert gam[:, nsig:].mean() < 0.002
    assert np.all(beta[gam == 0] == 0.0)
    assert abs(np.sqrt(sig).mean() - 1.0) < 0.05
    sm = eng.get_summaries()
    assert sm["sweeps"] == chains * nsw and sm["min_margin"] > 1e-9


def test_convenience_ctor_priors_match_oracle(oracle):
    """a0: the priors ba_set_priors_ctor1/2 assemble on the engine's own
    sufficient statistics against the oracle's restatement of
    BregVsSampler.cpp:48-142 (itself pinned by tests/golden/ssvs_ctors.npz)."""
    import boom_amd
    X, y, _ = regression_data(700, 24, 5, seed=33)
    eng = boom_amd.Engine(2, seed=1)
    eng.build_suf_from_xy(X, y)
    suf = _engine_suf(eng)
    for args in [(1.5, 0.6, 3.0, True), (0.7, 0.3, 40.0, False)]:
        eng.set_priors_ctor1(*args)
        got = eng.get_priors()
        want = oracle.prior_ctor1(suf, *args)
        assert np.array_equal(got["b"], want["b"])
        assert np.array_equal(got["pi"], want["pi"])
        assert relerr(got["ominv"], want["ominv"], 1e-12) < 1e-14
        assert abs(got["df"] - want["df"]) <= 1e-15 * want["df"]
        ss_want = want["df"] * want["sigma_guess"] ** 2
        assert abs(got["ss"] - ss_want) <= 1e-14 * ss_want
    for args in [(2.0, 1.3, 1.0, 0.5, 0.2, True), (1.0, 0.8, 2.5, 0.0, 0.1, False),
                 (1.0, 0.8, 2.5, 1.0, 0.1, True)]:
        eng.set_priors_ctor2(*args)
        got = eng.get_priors()
        want = oracle.prior_ctor2(suf, *args)
        assert np.array_equal(got["b"], want["b"])
        assert np.array_equal(got["pi"], want["pi"])
        assert relerr(got["ominv"], want["ominv"], 1e-12) < 1e-14
        assert abs(got["df"] - want["df"]) <= 1e-15 * want["df"]
        ss_want = want["df"] * want["sigma_guess"] ** 2
        assert abs(got["ss"] - ss_want) <= 1e-14 * ss_want
    # and the chain they drive is the oracle's chain with those priors
    eng.set_priors_ctor1(1.5, 0.6, 3.0, True)
    want = oracle.prior_ctor1(suf, 1.5, 0.6, 3.0, True)
    g0 = np.zeros(24, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.sweep(30)
    gam, beta, sig = eng.get_states()
    for c in range(2):
        o = oracle.ssvs_run(suf, want, ssvs_options(), ("philox", 1, c), g0, 30)
        assert np.array_equal(gam[c], o["gamma"][-1])
        assert relerr(beta[c], o["beta"][-1]) < RTOL


def test_row_sharded_suf_build_matches_single_shot(oracle):
    """config-4 data path on one device: the rows in three uneven shards, a
    partial block each (ba_suf_partial_device, the MFMA syrk on the shard),
    summed as the all-reduce would, installed with ba_set_suf_from_block_device:
    the statistics equal the single-shot build's to rounding, and the chains run
    on them are the oracle's chains on the same statistics"""
    import boom_amd
    import torch
    from boom_amd import dist as bd
    n, p = 5000, 96
    X, y, _ = regression_data(n, p, 7, seed=61)
    whole = boom_amd.Engine(4, seed=3)
    whole.build_suf_from_xy(X, y)
    ref = whole.get_suf()
    eng = boom_amd.Engine(4, seed=3)
    total = torch.zeros(bd.suf_block_size(p), dtype=torch.float64, device="cuda")
    bounds = [0, 1700, 1701, n]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        Xs = torch.from_numpy(np.ascontiguousarray(X[lo:hi].T)).cuda()   # column-major shard
        ys = torch.from_numpy(np.ascontiguousarray(y[lo:hi])).cuda()
        blk = torch.empty_like(total)
        eng.suf_partial_device(hi - lo, p, Xs.data_ptr(), ys.data_ptr(), blk.data_ptr())
        total += blk
    torch.cuda.synchronize()
    eng.set_suf_from_block_device(n, p, total.data_ptr())
    got = eng.get_suf()
    assert np.max(np.abs(got["xtx"] - ref["xtx"])) < 1e-12 * np.abs(ref["xtx"]).max()
    assert np.array_equal(got["xtx"], got["xtx"].T)
    assert relerr(got["xty"], ref["xty"], 1e-6) < 1e-12
    assert abs(got["yty"] - ref["yty"]) < 1e-12 * ref["yty"]
    assert abs(got["ybar"] - ref["ybar"]) < 1e-13 and got["n"] == n
    assert relerr(got["xbar"], ref["xbar"], 1e-6) < 1e-12
    suf = _engine_suf(eng)
    prior = spike_slab_prior(suf, 7)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.sweep(40)
    gam, beta, sig = eng.get_states()
    for c in (0, 3):
        o = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", 3, c), g0, 40)
        assert np.array_equal(gam[c], o["gamma"][-1])
        assert relerr(beta[c], o["beta"][-1]) < RTOL


def _c2_engine(chains, seed, X, y, nsig):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed)
    eng.build_suf_from_xy(X, y)
    suf = _engine_suf(eng)
    prior = spike_slab_prior(suf, nsig)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    return eng, suf, prior


def test_c2_every_chain_of_the_headline_against_the_oracle(oracle):
    """VERDICT r5, weak 1a / task 6a + 6c: ALL 1024 chains of BASELINE configs[1] against the
    oracle (BregVsSampler::draw, BregVsSampler.cpp:252-261), not four of them -- at 40 and at
    100 sweeps from the start (the models grow from the intercept to ~16 variables on the way:
    table fills, forks, both slots), once as ONE launch per checkpoint and once in the
    HEADLINE's mode: consecutive launches that overlap on two streams and hand the chains
    over one by one, with nothing between them.  gamma identical, beta / sigma^2 to 1e-8."""
    import os
    n, p, nsig, chains, seed = 10000, 512, 16, 1024, 8675309
    X, y, _ = regression_data(n, p, nsig, seed=8675309)
    one, suf, prior = _c2_engine(chains, seed, X, y, nsig)
    lap, _, _ = _c2_engine(chains, seed, X, y, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    one.set_state(g0)
    lap.set_state(g0)
    threads = max(1, len(os.sched_getaffinity(0)))
    done = 0
    for upto in (40, 100):
        one.sweep(upto - done)
        for _ in range((upto - done) // 10):      # launches of 10 sweeps, nothing in between
            lap.sweep(10, sync=False)
        lap.sync()
        done = upto
        o = oracle.run_chains(suf, prior, ssvs_options(), seed, chains, upto, threads, g0)
        assert o["status"] == 0
        for tag, eng in (("one launch", one), ("overlapped launches", lap)):
            gam, beta, sig = eng.get_states()
            bad = np.where((gam != o["gamma"]).any(1))[0]
            assert len(bad) == 0, (tag, upto, "chains whose inclusion indicators differ", bad[:10])
            assert relerr(beta, o["beta"]) < RTOL, (tag, upto)
            assert np.max(np.abs(sig - o["sigsq"]) / o["sigsq"]) < RTOL, (tag, upto)
    assert 14.0 < o["gamma"].sum(1).mean() < 18.0    # (the chains did get to the benchmark's models)
    one.close()
    lap.close()


def test_c3_sixty_four_chains_of_the_benchmark_shape(oracle):
    """task 6b: BASELINE configs[2] (T=2000, p=100, 1024 chains in the persistent round
    kernel), every sixteenth chain -- 64 of them, one per X'e tile position class -- against
    the oracle round by round (StateSpacePosteriorSampler::draw, :42-64)."""
    _ss_compare(oracle, T=2000, p=100, nsig=5, chains=1024, check=list(range(7, 1024, 16)), nsw=6,
                seed=4, data_seed=8675309)
