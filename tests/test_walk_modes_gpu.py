"""The sweep kernel has several equivalent ways of walking a sweep (batch /
table look-ups, forked quiet sweeps, 1 or 2 wavefronts per chain, one long or
many short launches).  They must produce the SAME chains bit for bit -- same
decisions, same arithmetic -- so the oracle parity of the default path
(test_ssvs_gpu.py) carries over to every one of them.  Through the C-ABI."""
import os

import numpy as np
import pytest

from cases import regression_data, spike_slab_prior, suf_from_xy
from oracle_lib import ssvs_options
from test_ssvs_gpu import make_engine

pytestmark = pytest.mark.gpu


def _cases():
    out = {}
    # plain: p spans several 64-proposal rounds
    X, y, _ = regression_data(600, 130, 7, seed=5)
    suf = suf_from_xy(X, y)
    out["plain_p130"] = (suf, spike_slab_prior(suf, 7), ssvs_options())
    # swap move with real candidates (collinear columns), low threshold
    X, y, _ = regression_data(400, 40, 4, seed=6, collinear=[1, 7, 9, 20])
    suf = suf_from_xy(X, y)
    out["collinear_swap"] = (suf, spike_slab_prior(suf, 4), ssvs_options(swap_threshold=0.5))
    # non-zero prior means (exact-path stops), model-size cap, few flips per sweep
    X, y, _ = regression_data(300, 70, 5, seed=7)
    suf = suf_from_xy(X, y)
    pm = np.zeros(70)
    pm[[0, 3, 11, 40]] = [0.5, -0.2, 0.1, 0.3]
    out["general"] = (suf, spike_slab_prior(suf, 5, prior_mean=pm),
                      ssvs_options(max_model_size=9, max_flips=50))
    return out


CASES = _cases()


def _run(case, waves, policy, launches, chains=12, seed=77):
    suf, prior, opts = CASES[case]
    p = len(suf["xty"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng = make_engine(chains, seed, suf=suf, prior=prior, opts=opts, g0=g0,
                      tuning=dict(waves_per_chain=waves, walk_policy=policy))
    for n in launches:
        eng.sweep(n)
    gam, beta, sig = eng.get_states()
    sm = eng.get_summaries()
    return gam, beta, sig, sm


@pytest.mark.parametrize("case", sorted(CASES))
def test_all_walks_give_the_same_chains(case):
    ref = _run(case, waves=1, policy=0, launches=[60])
    for waves, policy, launches in [(1, 2, [60]), (2, 0, [60]), (2, 1, [60]), (2, 2, [60]),
                                    (2, 3, [60]), (2, 2, [1] * 7 + [13, 40]),
                                    (1, 1, [20, 20, 20])]:
        got = _run(case, waves, policy, launches)
        tag = (case, waves, policy, launches)
        assert np.array_equal(ref[0], got[0]), tag
        assert np.array_equal(ref[1], got[1]), tag
        assert np.array_equal(ref[2], got[2]), tag
        assert ref[3]["sweeps"] == got[3]["sweeps"] and ref[3]["accepts"] == got[3]["accepts"], tag
        assert ref[3]["proposals"] == got[3]["proposals"], tag
        assert np.array_equal(ref[3]["inclusion_count"], got[3]["inclusion_count"]), tag
        assert np.allclose(ref[3]["beta_sum"], got[3]["beta_sum"], rtol=1e-12, atol=1e-12), tag


def test_tables_are_dropped_when_anything_else_is_called():
    """the per-chain proposal tables survive from one ba_sweep to the next; any
    other call in between (here: new sufficient statistics) must drop them"""
    suf, prior, opts = CASES["plain_p130"]
    p = len(suf["xty"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    X2, y2, _ = regression_data(600, 130, 7, seed=99)
    suf2 = suf_from_xy(X2, y2)
    a = make_engine(8, 5, suf=suf, prior=prior, opts=opts, g0=g0,
                    tuning=dict(walk_policy=2))
    a.sweep(30)
    a.upload_suf(suf2["xtx"], suf2["xty"], suf2["yty"], suf2["n"],
                 suf2["sumy"] / suf2["n"], suf2["xsum"] / suf2["n"])
    a.sweep(30)
    ga, ba_, sa = a.get_states()
    b = make_engine(8, 5, suf=suf, prior=prior, opts=opts, g0=g0,
                    tuning=dict(walk_policy=0))
    b.sweep(30)
    b.upload_suf(suf2["xtx"], suf2["xty"], suf2["yty"], suf2["n"],
                 suf2["sumy"] / suf2["n"], suf2["xsum"] / suf2["n"])
    b.sweep(30)
    gb, bb, sb = b.get_states()
    assert np.array_equal(ga, gb) and np.array_equal(ba_, bb) and np.array_equal(sa, sb)


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("waves,policy", [(2, 1), (2, 2), (1, 2)])
def test_every_draw_of_one_long_launch_matches_the_oracle(oracle, case, waves, policy):
    """ONE launch of many sweeps (table look-ups, forked quiet sweeps, kept
    model blocks all in play), every sweep's draw recorded on the device and
    compared with the oracle's draw of the same sweep: gamma bit-exact, beta and
    sigma^2 within the stated fp64 tolerance."""
    suf, prior, opts = CASES[case]
    p = len(suf["xty"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 6, 2024, 80
    eng = make_engine(chains, seed, suf=suf, prior=prior, opts=opts, g0=g0,
                      tuning=dict(waves_per_chain=waves, walk_policy=policy))
    eng.enable_draws(nsw)
    eng.sweep(nsw)
    draws = [eng.get_draws(c, nsw) for c in range(chains)]
    for c in range(chains):
        o = oracle.ssvs_run(suf, prior, opts, ("philox", seed, c), g0, nsw)
        assert o["status"] == 0
        gam, beta, sig = draws[c]
        for s in range(nsw):
            assert np.array_equal(gam[s], o["gamma"][s]), (case, c, s)
            err = np.max(np.abs(beta[s] - o["beta"][s]) / np.maximum(np.abs(o["beta"][s]), 1e-3))
            assert err < 1e-8, (case, c, s, err)
            assert abs(sig[s] - o["sigsq"][s]) < 1e-8 * sig[s], (case, c, s)


def test_predict_from_the_record():
    """lm_spike.predict (spikeslab.py:530-546) is coefficient_draws[burn:, :] @
    predictors.T on the draws the caller kept; ba_predict forms it on the device from
    its own record, for every chain."""
    import boom_amd
    X, y, _ = regression_data(500, 40, 5, seed=2)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, 5)
    eng = boom_amd.Engine(6, seed=9)
    eng.build_suf_from_xy(X, y)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(40, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    nsw, burn = 50, 12
    eng.enable_draws(nsw)
    eng.sweep(nsw)
    newX = np.random.Generator(np.random.PCG64(3)).standard_normal((33, 40))
    got = eng.predict(newX, burn, nsw - burn)
    assert got.shape == (6, nsw - burn, 33)
    for c in (0, 5):
        _, beta, _ = eng.get_draws(c, nsw)
        want = beta[burn:] @ newX.T
        assert np.max(np.abs(got[c] - want)) < 1e-12 * max(1.0, np.abs(want).max())


def test_more_chains_than_the_machine_holds_go_out_in_groups():
    """2304 chains on a GPU that holds 1024 workgroups of this kernel: the chains go out in
    groups of 1024 on alternating streams (engine.hip, sweep_impl) -- calls left to overlap,
    calls with readers and a mutator in between, and three engines of 768 chains with the
    matching chain offsets all give the same chains, bit for bit"""
    import boom_amd
    from cases import regression_data, spike_slab_prior
    X, y, _ = regression_data(3000, 96, 6, seed=12)
    def make(chains, offset=0):
        e = boom_amd.Engine(chains, seed=21, chain_offset=offset)
        e.build_suf_from_xy(X, y)
        s = e.get_suf()
        suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
        pr = spike_slab_prior(suf, 6)
        e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
        g0 = np.zeros(96, np.uint8)
        g0[0] = 1
        e.set_state(g0)
        return e
    a, b = make(2304), make(2304)
    for it in range(6):
        a.sweep(25, sync=False)            # left to overlap
        b.sweep(25)                        # one call at a time
        if it == 2:
            a.set_options(max_flips=40)
            b.set_options(max_flips=40)
        if it == 3:
            assert np.array_equal(a.get_state(1500)[1], b.get_state(1500)[1])
    a.sync()
    sa, sb = a.get_states(), b.get_states()
    for u, v in zip(sa, sb):
        assert np.array_equal(u, v)
    parts = []
    for k in range(3):
        e = make(768, 768 * k)
        for it in range(6):
            e.sweep(25)
            if it == 2:
                e.set_options(max_flips=40)
        parts.append(e.get_states())
    for i in range(3):
        assert np.array_equal(np.concatenate([q[i] for q in parts]), sa[i])
