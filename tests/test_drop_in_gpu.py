"""The drop-in calling pattern and the call-sequence hazards around it.

* ba_draw_next(): one draw per call served from a device-side record, for
  EVERY chain, bitwise what one ba_sweep(1) per iteration gives, whatever the
  look-ahead length and whatever is called in between.
* ba_set_state after an asynchronous sweep, ba_log_model_prob after
  a fixed-precision SpikeSlab sweep, ba_seed on the SpikeSlab / state-space
  streams, chain indices on a shard with chain_offset != 0.
"""
import os

import numpy as np
import pytest

from cases import bsts_priors, regression_data, spike_slab_prior, state_space_data, suf_from_xy
from oracle_lib import ssvs_options
from test_ssvs_gpu import make_engine

pytestmark = pytest.mark.gpu


def _case(p=40, nsig=5, seed=3, n=400):
    X, y, _ = regression_data(n, p, nsig, seed=seed)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    return suf, prior, g0


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("lookahead", [2, 7, 64])
def test_draw_next_serves_every_chain_the_per_call_draws(lookahead):
    suf, prior, g0 = _case()
    chains, niter = 9, 45
    a = make_engine(chains, 5, suf=suf, prior=prior, g0=g0, chain_offset=100)
    b = make_engine(chains, 5, suf=suf, prior=prior, g0=g0, chain_offset=100)
    b.set_lookahead(lookahead)
    for it in range(niter):
        a.sweep(1)
        b.draw_next()
        assert _same(a.get_states(), b.get_states()), it
        for c in (0, chains - 1):
            ga, ba_, sa = a.get_state(c)
            gb, bb, sb = b.get_state(c)
            assert np.array_equal(ga, gb) and np.array_equal(ba_, bb) and sa == sb
            assert a.logpri(c) == b.logpri(c)


def test_draw_next_rewinds_when_anything_else_is_called():
    """setters, sweeps and state writes between served draws see the chain where
    the caller has seen it, not at the end of the look-ahead batch"""
    suf, prior, g0 = _case()
    chains = 6
    a = make_engine(chains, 8, suf=suf, prior=prior, g0=g0)
    b = make_engine(chains, 8, suf=suf, prior=prior, g0=g0)
    b.set_lookahead(10)
    a.reset_summaries()
    b.reset_summaries()
    for _ in range(13):          # 3 draws into the second batch
        a.sweep(1)
        b.draw_next()
    a.set_options(max_flips=17)
    b.set_options(max_flips=17)  # rewinds 7 unserved draws, replays 3
    assert _same(a.get_states(), b.get_states())
    sa, sb = a.get_summaries(), b.get_summaries()
    assert sa["sweeps"] == sb["sweeps"] == chains * 13
    assert np.array_equal(sa["inclusion_count"], sb["inclusion_count"])
    for _ in range(4):
        a.sweep(1)
        b.draw_next()
    assert _same(a.get_states(), b.get_states())
    a.sweep(6)
    b.sweep(6)                   # plain sweeps continue after the last SERVED draw
    assert _same(a.get_states(), b.get_states())
    for _ in range(3):
        a.sweep(1)
        b.draw_next()
    g1 = np.zeros(len(g0), np.uint8)
    g1[[0, 2, 5]] = 1
    beta1 = np.linspace(0, 1, len(g0)) * g1
    a.set_state(g1, beta1, 0.7, chain=2)
    b.set_state(g1, beta1, 0.7, chain=2)
    for _ in range(12):
        a.sweep(1)
        b.draw_next()
    assert _same(a.get_states(), b.get_states())
    b.set_lookahead(1)
    a.sweep(1)
    b.draw_next()
    assert _same(a.get_states(), b.get_states())


def test_set_state_after_asynchronous_sweeps():
    """ba_set_state must order against launches in flight"""
    suf, prior, g0 = _case(p=130, nsig=7, seed=5)
    g1 = np.zeros(130, np.uint8)
    g1[[0, 3, 9]] = 1
    a = make_engine(64, 2, suf=suf, prior=prior, g0=g0)
    b = make_engine(64, 2, suf=suf, prior=prior, g0=g0)
    for _ in range(4):
        a.sweep(50, sync=False)
    a.set_state(g1, None, 2.0)      # no explicit sync by the caller
    a.sweep(5)
    b.sweep(200)
    b.set_state(g1, None, 2.0)
    b.sweep(5)
    assert _same(a.get_states(), b.get_states())
    # one chain only, the others keep running from where they were
    a.sweep(30, sync=False)
    a.set_state(g0, None, 1.0, chain=7)
    a.sweep(5)
    b.sweep(30)
    b.set_state(g0, None, 1.0, chain=7)
    b.sweep(5)
    assert _same(a.get_states(), b.get_states())


def test_log_model_prob_after_fixed_precision_spike_slab_sweep(oracle):
    """a SpikeSlabSampler launch with a slab precision that
    does not scale with sigma^2 leaves Omega^-1 + XtX / sigma^2 on the device;
    BregVsSampler::log_model_prob must not be computed from it."""
    suf, prior, g0 = _case(p=20, nsig=4, seed=9)
    eng = make_engine(3, 4, suf=suf, prior=prior, g0=g0)
    rng = np.random.Generator(np.random.PCG64(1))
    gammas = (rng.random((16, 20)) < 0.3).astype(np.uint8)
    gammas[:, 0] = 1
    want = oracle.log_model_prob(suf, prior, gammas)
    assert np.max(np.abs(eng.log_model_prob(gammas) - want)) < 1e-9 * np.abs(want).max()
    eng.sss_set_slab(prior["b"], prior["ominv"], scales_with_sigsq=False)
    eng.set_sigsq(0.37)
    eng.sss_sweep(3)
    got = eng.log_model_prob(gammas)
    assert np.max(np.abs(got - want)) < 1e-9 * np.abs(want).max()
    eng.sweep(2)                    # and a BregVs sweep after it runs on the right V again
    assert np.max(np.abs(eng.log_model_prob(gammas) - want)) < 1e-9 * np.abs(want).max()


def test_reseeding_restarts_every_stream():
    """seed-then-run equals a fresh engine with that seed, for
    the SpikeSlab stream and the three state-space streams as well."""
    import boom_amd
    suf, prior, g0 = _case(p=24, nsig=4, seed=12)

    def sss(eng):
        eng.sss_set_slab(prior["b"], prior["ominv"], scales_with_sigsq=True)
        eng.set_state(g0)
        eng.set_sigsq(1.3)
        eng.sss_sweep(8)
        return eng.get_states()
    a = make_engine(4, 111, suf=suf, prior=prior, g0=g0)
    sss(a)
    a.seed(222)
    b = make_engine(4, 222, suf=suf, prior=prior, g0=g0)
    assert _same(sss(a), sss(b))

    T, p = 150, 6
    X, y, _, obs = state_space_data(T, p, 2, seed=7)
    pr, ss, sig_up = bsts_priors(X, y, 2)

    def mk(seed):
        eng = boom_amd.Engine(3, seed=seed)
        eng.ss_set_data(y, X, obs)
        eng.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"],
                       sigma_upper_limit=sig_up)
        eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"],
                               ss["level_sigma_upper_limit"], ss["initial_state_mean"],
                               ss["initial_state_variance"], ss["initial_level_sigma"])
        eng.set_state(np.zeros(p, np.uint8))
        return eng

    def reset(eng):
        eng.set_state(np.zeros(p, np.uint8))
        eng.ss_set_level_sigsq(1.0)
    a = mk(5)
    a.ss_sweep(6)
    a.seed(9)
    a.ss_set_data(y, X, obs)        # back to the data's own sufficient statistics (what changed is the key)
    reset(a)
    b = mk(9)
    reset(b)
    a.ss_sweep(6)
    b.ss_sweep(6)
    assert _same(a.get_states(), b.get_states())
    for c in range(3):
        sa, sb = a.ss_get_state(c), b.ss_get_state(c)
        assert np.array_equal(sa["state"], sb["state"]) and sa["level_sigsq"] == sb["level_sigsq"]


def test_log_model_prob_is_refused_in_state_space_mode():
    import boom_amd
    T, p = 100, 5
    X, y, _, obs = state_space_data(T, p, 2, seed=7)
    pr, ss, sig_up = bsts_priors(X, y, 2)
    eng = boom_amd.Engine(2, seed=1)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"],
                   sigma_upper_limit=sig_up)
    with pytest.raises(boom_amd.BoomAmdError):
        eng.log_model_prob(np.ones((1, p), np.uint8))


def test_log_model_prob_of_models_beyond_64_variables(oracle):
    """BregVsSampler::log_model_prob has no size limit (BregVsSampler.cpp:216-239): vectors
    with more included variables than the LDS kernel holds go through the large-model
    build, mixed freely with small ones in one call."""
    suf, prior, g0 = _case(p=300, nsig=6, seed=13, n=1500)
    eng = make_engine(2, 4, suf=suf, prior=prior, g0=g0)
    rng = np.random.Generator(np.random.PCG64(5))
    gammas = np.zeros((9, 300), np.uint8)
    for row, k in enumerate([3, 65, 100, 64, 129, 200, 0, 300, 70]):
        gammas[row, rng.choice(300, k, replace=False)] = 1
    gammas[:8, 0] = 1             # (the prior forces the intercept in: the last row stays illegal)
    gammas[8, 0] = 0
    want = oracle.log_model_prob(suf, prior, gammas)
    got = eng.log_model_prob(gammas)
    fin = np.isfinite(want)
    assert fin[:8].all() and not fin[8] and np.array_equal(got[~fin], want[~fin])
    assert np.max(np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))) < 1e-9
    # the chains' own models are untouched by it
    eng.sweep(3)
    ref = make_engine(2, 4, suf=suf, prior=prior, g0=g0)
    ref.sweep(3)
    assert _same(eng.get_states(), ref.get_states())


def test_overlapping_sweep_launches_hand_chains_over(oracle):
    """Consecutive ba_sweep calls with nothing in between run on two streams and hand the
    chains over one by one (a workgroup of the next launch takes a chain the current launch
    is done with): the draws are those of one launch at a time, for every chain, and any
    other call in between (a state read, a prior change) joins the pipeline first."""
    import boom_amd
    n, p, nsig, chains = 2000, 64, 6, 96
    X, y, _ = regression_data(n, p, nsig, seed=17)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1

    def fresh():
        return make_engine(chains, 123, suf=suf, prior=prior, g0=g0)
    a, b = fresh(), fresh()
    plan = [7, 1, 30, 2, 2, 19, 64, 1, 1, 40]
    for i, k in enumerate(plan):
        a.sweep(k, sync=False)            # (no sync: the launches overlap)
        if i == 5:
            ga, _, _ = a.get_states()     # a reader in the middle joins the pipeline
            a.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    a.sync()
    for i, k in enumerate(plan):
        b.sweep(k, sync=True)
        if i == 5:
            gb, _, _ = b.get_states()
            b.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    assert np.array_equal(ga, gb)
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)
    sa, sb = a.get_summaries(), b.get_summaries()
    assert sa["sweeps"] == sb["sweeps"] and sa["sweeps"] > 0
    assert np.array_equal(sa["inclusion_count"], sb["inclusion_count"])
    # ... and the chains are the oracle's
    o = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", 123, chains - 1), g0, sum(plan))
    gam, beta, sig = a.get_states()
    assert np.array_equal(gam[chains - 1], o["gamma"][-1])
    assert abs(sig[chains - 1] - o["sigsq"][-1]) < 1e-8 * sig[chains - 1]


def test_a_prior_change_between_two_unsynced_sweeps_joins_the_pipeline():
    """ba_sweep; ba_sweep; ba_set_spike; ba_sweep with nothing synced in between.
    The second launch is still running on the other stream when the setter arrives: it must
    be behind the main stream before the shared prior arrays are overwritten, or it reads
    half of the new values.  Same chains as the synced sequence, bit for bit."""
    suf, prior, g0 = _case(p=64, nsig=6, seed=17, n=2000)
    chains = 96
    a = make_engine(chains, 123, suf=suf, prior=prior, g0=g0)
    b = make_engine(chains, 123, suf=suf, prior=prior, g0=g0)
    pi2 = np.clip(prior["pi"] * 3.0, 0.0, 1.0)
    for rep in range(3):
        a.sweep(40, sync=False)
        a.sweep(40, sync=False)
        a.set_spike(pi2 if rep % 2 == 0 else prior["pi"])        # (no sync: the launches overlap)
        a.sweep(25, sync=False)
        a.set_options(max_flips=20 if rep == 1 else -1)
        a.sweep(10, sync=False)
    a.sync()
    for rep in range(3):
        b.sweep(40)
        b.sweep(40)
        b.set_spike(pi2 if rep % 2 == 0 else prior["pi"])
        b.sweep(25)
        b.set_options(max_flips=20 if rep == 1 else -1)
        b.sweep(10)
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)
    sa, sb = a.get_summaries(), b.get_summaries()
    assert np.array_equal(sa["inclusion_count"], sb["inclusion_count"])


def test_overlapping_lookahead_batches_serve_the_per_call_draws():
    """the callers' loop as they write it -- draw_next(); get_state(0) -- over many batches:
    the batch after the one being served is already running (handing chains over launch to
    launch), the served draws are those of one launch per call, for chain 0 at every
    iteration and for every chain wherever one looks; a mutator in the middle of a batch
    (with the next batch in flight) and one exactly at a batch's end rewind correctly"""
    suf, prior, g0 = _case(p=48, nsig=6, seed=5, n=600)
    chains, L = 40, 16
    a = make_engine(chains, 21, suf=suf, prior=prior, g0=g0)
    b = make_engine(chains, 21, suf=suf, prior=prior, g0=g0)
    b.set_lookahead(L)
    for it in range(1, 5 * L + 4):
        a.sweep(1)
        b.draw_next()
        ga, ba_, sa = a.get_state(0)
        gb, bb, sb = b.get_state(0)
        assert np.array_equal(ga, gb) and np.array_equal(ba_, bb) and sa == sb, it
        if it % 13 == 0:
            assert _same(a.get_state(chains - 1), b.get_state(chains - 1)), it
        if it == 2 * L + 5:          # mid-batch, next batch in flight
            a.set_options(max_flips=30)
            b.set_options(max_flips=30)
        if it == 4 * L:              # the batch's last draw served, next batch in flight
            assert _same(a.get_states(), b.get_states())
            a.set_options(max_flips=-1)
            b.set_options(max_flips=-1)
    assert _same(a.get_states(), b.get_states())
    a.sync()
    b.sync()
    for x in (a, b):
        x.set_options(max_flips=-1)  # (a mutator: the look-ahead engine drops what it ran ahead)
    sa, sb = a.get_summaries(), b.get_summaries()
    assert sa["sweeps"] == sb["sweeps"]
    assert np.array_equal(sa["inclusion_count"], sb["inclusion_count"])


def test_overlapping_lookahead_batches_survive_a_capacity_stop():
    """chains that outgrow the launch's capacity inside an overlapped batch: the batch is
    run again the way batches ran before (escalation included), with the same draws"""
    suf, prior, g0 = _case(p=64, nsig=28, seed=9, n=800)
    chains, L = 12, 8
    a = make_engine(chains, 3, suf=suf, prior=prior, g0=g0, tuning=dict(kcap_start=16))
    b = make_engine(chains, 3, suf=suf, prior=prior, g0=g0, tuning=dict(kcap_start=16))
    b.set_lookahead(L)
    for it in range(6 * L):
        a.sweep(1)
        b.draw_next()
        assert _same(a.get_state(0), b.get_state(0)), it
        assert _same(a.get_state(chains - 1), b.get_state(chains - 1)), it
    assert _same(a.get_states(), b.get_states())
    gam, _, _ = a.get_states()
    assert gam.sum(axis=1).max() > 16       # the models did outgrow the first capacity


def test_twice_as_many_chains_as_fit_go_out_as_two_launches(oracle):
    """exactly 2 x (4 chains per CU) chains: the engine issues the sweep as two launches of
    resident size (one multi-round launch starts a few workgroups a round late on this
    machine); chains of both halves are the oracle's, and a second call continues them"""
    import boom_amd
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    chains = 8 * cus
    suf, prior, g0 = _case(p=24, nsig=4, seed=2, n=300)
    eng = make_engine(chains, 77, suf=suf, prior=prior, g0=g0)
    eng.sweep(12)
    eng.sweep(13)
    gam, beta, sig = eng.get_states()
    for c in (0, chains // 2 - 1, chains // 2, chains - 1):
        o = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", 77, c), g0, 25)
        assert np.array_equal(gam[c], o["gamma"][-1]), c
        assert abs(sig[c] - o["sigsq"][-1]) < 1e-8 * sig[c], c
    assert eng.get_summaries()["sweeps"] == chains * 25
