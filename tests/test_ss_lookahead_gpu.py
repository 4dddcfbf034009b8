"""The look-ahead on the bsts path (ba_ss_set_lookahead / ba_ss_draw_next): the callers'
loop -- draw_next(); read chain 0 -- served from the device's record of batches enqueued
ahead, against one ba_ss_sweep(1) per iteration.  The look-ahead must be unobservable:
every accessor sees the draw being served, for every chain; anything that is not in the
record (a mutator, a sufficient statistic, another chain's state path, a forecast, a plain
ba_ss_sweep) first puts the chains back at that draw.  Equality is exact: the same kernels
run the same rounds on the same stream positions.
(StateSpacePosteriorSampler.cpp:42-64, Interfaces/R/bsts/src/bsts.cc:82-119.)"""
import numpy as np
import pytest

from cases import bsts_priors, general_data, general_spec, state_space_data
from test_state_space_gpu import make_engine as make_level_engine
from test_structural_general_gpu import make_engine as make_general_engine

pytestmark = pytest.mark.gpu


def same_level_draw(a, b, chains):
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)
    for c in chains:
        sa, sb = a.ss_get_state(c), b.ss_get_state(c)
        assert sa["level_sigsq"] == sb["level_sigsq"], c
        assert np.array_equal(sa["state"], sb["state"]), c


def test_local_level_draw_next_serves_the_per_call_draws():
    """chain 0 at every iteration, every chain's regression draw and level variance now
    and then, another chain's state path (not in the record: served by going back to the
    draw), mutators in the middle of a batch, at a batch's last draw and at its first"""
    T, p, chains, L = 150, 8, 12, 8
    X, y, _, obs = state_space_data(T, p, 3, seed=5, missing_frac=0.03)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    g0 = np.zeros(p, np.uint8)
    a = make_level_engine(chains, 31, y, X, obs, prior, ss, sig_up, g0)
    b = make_level_engine(chains, 31, y, X, obs, prior, ss, sig_up, g0)
    b.ss_set_lookahead(L)
    for it in range(1, 6 * L + 3):
        a.ss_sweep(1)
        b.ss_draw_next()
        ga, ba_, sa = a.get_state(0)
        gb, bb, sb = b.get_state(0)
        assert np.array_equal(ga, gb) and np.array_equal(ba_, bb) and sa == sb, it
        ua, ub = a.ss_get_state(0, suf=False), b.ss_get_state(0, suf=False)
        assert ua["level_sigsq"] == ub["level_sigsq"] and np.array_equal(ua["state"], ub["state"]), it
        assert a.logpri(0) == b.logpri(0), it
        if it % 5 == 0:                      # every chain, from the record
            for u, v in zip(a.get_states(), b.get_states()):
                assert np.array_equal(u, v), it
            c = chains - 1
            assert a.ss_get_state(c, state=False, suf=False)["level_sigsq"] == b.ss_get_state(c, state=False, suf=False)["level_sigsq"]
        if it == L + 3:                      # a state path that is not recorded
            same_level_draw(a, b, [chains - 1])
        if it == 2 * L + 4:                  # a mutator mid-batch, the next batch in flight
            a.set_options(max_flips=5)
            b.set_options(max_flips=5)
        if it == 4 * L:                      # ... at the batch's last draw
            a.ss_set_level_sigsq(0.3)
            b.ss_set_level_sigsq(0.3)
        if it == 5 * L + 1:                  # ... at a batch's first draw
            a.set_state(g0, chain=1)
            b.set_state(g0, chain=1)
    # the sufficient statistics are not in the record; a plain sweep goes on from the draw served last
    sa, sb = a.ss_get_state(2), b.ss_get_state(2)
    assert sa["level_n"] == sb["level_n"] and sa["level_sumsq"] == sb["level_sumsq"]
    a.ss_sweep(3)
    b.ss_sweep(3)
    same_level_draw(a, b, range(chains))
    # ... and so does a forecast
    b.ss_draw_next()
    a.ss_sweep(1)
    newX = np.random.Generator(np.random.PCG64(1)).standard_normal((6, p))
    assert np.array_equal(a.ss_forecast(newX), b.ss_forecast(newX))
    same_level_draw(a, b, range(chains))


def test_structural_draw_next_serves_the_per_call_draws():
    """the same for a general list of state models; two chains' state paths recorded"""
    T, p, chains, L = 90, 5, 7, 6
    desc = [("trend",), ("seasonal", 4, 3), ("ar", 2)]
    X, y, _, obs = general_data(T, p, 2, [(4, 3)], seed=12, missing_frac=0.03, ar_coef=[0.5])
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    g0 = np.zeros(p, np.uint8)
    a = make_general_engine(chains, 5, y, X, obs, prior, blocks, sig_up, g0)
    b = make_general_engine(chains, 5, y, X, obs, prior, blocks, sig_up, g0)
    b.ss_set_lookahead(L, chains=[0, 3])
    for it in range(1, 4 * L + 2):
        a.ss_sweep(1)
        b.ss_draw_next()
        for c in (0, 3):
            for u, v in zip(a.get_state(c), b.get_state(c)):
                assert np.array_equal(u, v), (it, c)
            assert np.array_equal(a.ss_get_state_draw(c), b.ss_get_state_draw(c)), (it, c)
        for k in range(len(blocks)):
            ma, mb = a.ss_get_state_model(chains - 1, k, suf=False), b.ss_get_state_model(chains - 1, k, suf=False)
            assert np.array_equal(ma["variances"], mb["variances"]), (it, k)
            if "phi" in ma:
                assert np.array_equal(ma["phi"], mb["phi"]), (it, k)
        if it == L + 2:       # sufficient statistics and an unrecorded state path: back to the draw
            ma, mb = a.ss_get_state_model(1, 2), b.ss_get_state_model(1, 2)
            assert np.array_equal(ma["xtx"], mb["xtx"]) and ma["n"] == mb["n"]
            assert np.array_equal(a.ss_get_state_draw(5), b.ss_get_state_draw(5))
    a.ss_sweep(2)
    b.ss_sweep(2)
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("desc", [[("trend",), ("seasonal", 7, 1)], [("level",), ("seasonal", 12, 1)],
                                  [("trend",), ("seasonal", 12, 1), ("ar", 1)], [("trend",)]])
def test_template_shapes_behind_the_look_ahead(desc):
    """the shapes that run the kernel compiled for them (ssm_template_kernel.hip), batches
    enqueued back to back while the other engine's launches run beside them.  (Round 4's
    stress runs, tools/ss_la_stress.py, found a race here that the per-call tests could not
    see: every thread of the state kernels draws the state models' variances from the stream
    position it reads, and thread 0 stored the new position without waiting for the other
    wave to have read the old one -- a wave that fell a draw behind under load drew another
    variance than its neighbour.)"""
    T, p, chains, L = 200, 8, 40, 6
    seas = [(d[1], d[2]) for d in desc if d[0] == "seasonal"]
    X, y, _, obs = general_data(T, p, 2, seas, seed=8, missing_frac=0.02,
                                ar_coef=[0.5] if any(d[0] == "ar" for d in desc) else None)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, desc)
    g0 = np.zeros(p, np.uint8)
    a = make_general_engine(chains, 7, y, X, obs, prior, blocks, sig_up, g0)
    b = make_general_engine(chains, 7, y, X, obs, prior, blocks, sig_up, g0)
    b.ss_set_lookahead(L)
    for it in range(60):
        a.ss_sweep(1)
        b.ss_draw_next()
        for u, v in zip(a.get_states(), b.get_states()):
            assert np.array_equal(u, v), it
        if it % 7 == 3:
            a.ss_sweep(2)
            b.ss_sweep(2)
        if it % 11 == 5:
            assert np.array_equal(a.ss_get_state_draw(17), b.ss_get_state_draw(17)), it


def test_a_capacity_stop_inside_a_batch():
    """chains that outgrow the sweep kernel's working capacity inside a look-ahead batch
    (40 signals from the empty model: 16 -> 32 -> 48): the batch is run again round by
    round with the stops dealt with where they happen -- the draws of one round per call"""
    T, p, chains, L = 120, 60, 6, 10
    X, y, _, obs = state_space_data(T, p, 40, seed=9)
    prior, ss, sig_up = bsts_priors(X, y, 40)
    g0 = np.zeros(p, np.uint8)
    a = make_level_engine(chains, 3, y, X, obs, prior, ss, sig_up, g0)
    b = make_level_engine(chains, 3, y, X, obs, prior, ss, sig_up, g0)
    b.ss_set_lookahead(L)
    for it in range(1, 3 * L + 1):
        a.ss_sweep(1)
        b.ss_draw_next()
        for u, v in zip(a.get_states(), b.get_states()):
            assert np.array_equal(u, v), it
        ua, ub = a.ss_get_state(0, suf=False), b.ss_get_state(0, suf=False)
        assert ua["level_sigsq"] == ub["level_sigsq"] and np.array_equal(ua["state"], ub["state"]), it
    # (every chain outgrew the first capacity, most of them the second: both stops happened inside
    # batches.  Round 6 re-rolled the state stream's numbers -- Box-Muller pairs -- and one chain
    # stands at 26 variables after these 30 rounds where all stood above 30 before.)
    k = a.get_states()[0].sum(axis=1)
    assert k.min() > 16 and k.max() > 32


@pytest.mark.parametrize("habit", ["statistics", "another chain's state", "a mutator"])
def test_callers_that_defeat_the_look_ahead(habit):
    """A caller's loop that after EVERY draw asks for something the record does not hold
    (sufficient statistics), reads the state path of a chain that is not recorded, or changes
    something: each such call puts the chains back at the draw being served.  The engine
    bounds what that costs -- the batches shrink (to one round per call) while it goes on and
    grow back when it stops, a chain whose state was asked for joins the record -- and the
    draws stay those of one round per call, through the shrinking, the tries with two rounds
    and the growing back."""
    T, p, chains, L = 180, 9, 10, 16
    X, y, _, obs = state_space_data(T, p, 3, seed=19, missing_frac=0.03)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    g0 = np.zeros(p, np.uint8)
    a = make_level_engine(chains, 3, y, X, obs, prior, ss, sig_up, g0)
    b = make_level_engine(chains, 3, y, X, obs, prior, ss, sig_up, g0)
    b.ss_set_lookahead(L)
    for it in range(90):
        a.ss_sweep(1)
        b.ss_draw_next()
        bad_habit = it < 45 or it >= 70          # (25 calm draws in the middle: the batches grow back)
        if bad_habit and habit == "statistics":
            u, v = a.ss_get_state(4), b.ss_get_state(4)
            assert u["level_sumsq"] == v["level_sumsq"] and np.array_equal(u["state"], v["state"]), it
        elif bad_habit and habit == "another chain's state":
            c = 2 + it % 3
            assert np.array_equal(a.ss_get_state(c, suf=False)["state"], b.ss_get_state(c, suf=False)["state"]), it
        elif bad_habit:
            mf = 2 + it % (p - 1)
            a.set_options(max_flips=mf)
            b.set_options(max_flips=mf)
        for u, v in zip(a.get_state(0), b.get_state(0)):
            assert np.array_equal(u, v), it
        assert np.array_equal(a.ss_get_state(0, suf=False)["state"], b.ss_get_state(0, suf=False)["state"]), it
    for u, v in zip(a.get_states(), b.get_states()):
        assert np.array_equal(u, v)
    a.close()
    b.close()
