"""The bsts local-level rounds as one persistent launch (ss_round_kernel.hip: every chain's
workgroup loops over the rounds of a call by itself, chains meet in X'e tiles formed in
arrival order) against the separate launches per round of rounds 1-4 -- the same chains, the
same stream positions: inclusion indicators identical, everything else within the fp64 bar
(the tile product sums a row's 128 steps in another order than the tiled GEMM did) -- and
against the oracle where the other state-space tests do not already go through it.
(StateSpacePosteriorSampler::draw, StateSpacePosteriorSampler.cpp:42-64.)
"""
import numpy as np
import pytest

from cases import bsts_priors, state_space_data
from oracle_lib import ssvs_options

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def make_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0, round_kernel, chain_offset=0):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed, chain_offset=chain_offset)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"],
                           ss["level_sigma_upper_limit"], ss["initial_state_mean"],
                           ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(g0)
    eng.ss_set_tuning(kernel=5 if round_kernel else 4)
    return eng


def same_chains(a, b, chains, what):
    ga, ba_, sa = a.get_states()
    gb, bb, sb = b.get_states()
    assert np.array_equal(ga, gb), what
    assert relerr(ba_, bb) < RTOL, what
    assert np.max(np.abs(sa - sb) / sb) < RTOL, what
    for c in chains:
        x, z = a.ss_get_state(c), b.ss_get_state(c)
        assert abs(x["level_sigsq"] - z["level_sigsq"]) <= RTOL * z["level_sigsq"], (what, c)
        assert np.max(np.abs(x["state"] - z["state"])) < RTOL * np.abs(z["state"]).max(), (what, c)
        assert abs(x["level_sumsq"] - z["level_sumsq"]) <= RTOL * max(z["level_sumsq"], 1e-300), (what, c)
        u, v = a.ss_get_chain_suf(c), b.ss_get_chain_suf(c)
        assert relerr(u["xty"], v["xty"], floor=1e-6 * np.abs(v["xty"]).max()) < RTOL, (what, c)
        assert abs(u["yty"] - v["yty"]) <= RTOL * v["yty"] and u["n"] == v["n"], (what, c)


@pytest.mark.parametrize("T,p,chains,missing", [(200, 8, 6, 0.05), (333, 20, 37, 0.0), (2000, 100, 160, 0.0),
                                                 (2048, 130, 33, 0.02), (17, 3, 16, 0.0)])
def test_round_kernel_equals_the_separate_launches(T, p, chains, missing):
    """calls of 1, 7 and 70 rounds (the last one is two launches: SS_ROUND_MAX_ROUNDS = 64),
    chain counts that are no multiple of the tile's 16, p beyond one pass of the tile product"""
    X, y, _, obs = state_space_data(T, p, 3, seed=5, missing_frac=missing)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    g0 = np.zeros(p, np.uint8)
    a = make_engine(chains, 77, y, X, obs, prior, ss, sig_up, g0, True)
    b = make_engine(chains, 77, y, X, obs, prior, ss, sig_up, g0, False)
    watch = sorted({0, 1, chains // 2, chains - 1})
    for n in (1, 7, 70, 2):
        a.ss_sweep(n)
        b.ss_sweep(n)
        same_chains(a, b, watch, "after %d more rounds" % n)
    a.close()
    b.close()


def test_round_kernel_beyond_the_32_bit_offsets_of_one_launch():
    """15 000 chains: a chain's work arrays are 147 KB, so the residual series of chain 14 563 and
    up lie more than 2^31 bytes behind the engine's first -- the tile product addresses them from
    the LAUNCH's first chain (from the engine's first it read zeros there, and other
    chains' series from 29 127 chains on).  The last chains against the separate launches."""
    T, p, chains = 40, 4, 15000
    X, y, _, obs = state_space_data(T, p, 2, seed=6)
    prior, ss, sig_up = bsts_priors(X, y, 2)
    g0 = np.zeros(p, np.uint8)
    a = make_engine(chains, 13, y, X, obs, prior, ss, sig_up, g0, True)
    b = make_engine(chains, 13, y, X, obs, prior, ss, sig_up, g0, False)
    for n in (1, 5):
        a.ss_sweep(n)
        b.ss_sweep(n)
        same_chains(a, b, [0, 14562, 14563, 14999], "after %d more rounds" % n)
    for c in (14600, 14999):   # bitwise: the same products in the same order
        assert np.array_equal(a.ss_get_chain_suf(c)["xty"], b.ss_get_chain_suf(c)["xty"]), c
    a.close()
    b.close()


def test_round_kernel_every_round_against_the_oracle(oracle):
    """one round per call and ten per call, chains 0 .. 19 draw by draw"""
    T, p, chains, nsw, seed = 300, 12, 20, 30, 41
    X, y, _, obs = state_space_data(T, p, 3, seed=9, missing_frac=0.03)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    ora = [oracle.ss_run(y, X, obs, prior, opts, ss, ("philox", seed, c), g0, nsw) for c in range(chains)]
    for per_call in (1, 10):
        eng = make_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0, True)
        for s in range(per_call - 1, nsw, per_call):
            eng.ss_sweep(per_call)
            gam, beta, sig = eng.get_states()
            for c in range(chains):
                o = ora[c]
                assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
                assert relerr(beta[c], o["beta"][s]) < RTOL, (c, s)
                assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], (c, s)
                st = eng.ss_get_state(c)
                assert abs(st["level_sigsq"] - o["level_sigsq"][s]) < RTOL * st["level_sigsq"]
                assert np.max(np.abs(st["state"] - o["state"][s])) < 1e-8 * np.abs(o["state"][s]).max()
        eng.close()


def test_round_kernel_capacity_stop_and_catch_up():
    """Seven signals among 40 regressors from the empty model at launch capacity 16: chains
    outgrow the capacity inside a call, sit the rest of it out (their rounds booked), are
    caught up one (sweep, state draw) pair at a time with the larger capacity -- the same
    draws as the separate launches make."""
    T, p, chains = 400, 40, 48
    X, y, _, obs = state_space_data(T, p, 24, seed=3)
    prior, ss, sig_up = bsts_priors(X, y, 24)
    g0 = np.zeros(p, np.uint8)
    a = make_engine(chains, 5, y, X, obs, prior, ss, sig_up, g0, True)
    b = make_engine(chains, 5, y, X, obs, prior, ss, sig_up, g0, False)
    for n in (40, 40, 5):
        a.ss_sweep(n)
        b.ss_sweep(n)
        same_chains(a, b, [0, 17, 47], "after %d more rounds" % n)
    assert a.get_states()[0].sum(1).max() > 16
    a.close()
    b.close()


def test_two_engines_rounds_side_by_side():
    """Two engines' persistent launches in flight together (each engine's launches go out in
    groups of chains that are co-resident beside the other's; a tile whose members are not
    all running is closed by its first member after 1 ms): each equals its own
    separate-launch twin."""
    T, p, chains = 500, 16, 1024
    X, y, _, obs = state_space_data(T, p, 3, seed=8)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    g0 = np.zeros(p, np.uint8)
    a = make_engine(chains, 1, y, X, obs, prior, ss, sig_up, g0, True)
    b = make_engine(chains, 2, y, X, obs, prior, ss, sig_up, g0, True)
    for _ in range(3):
        a.ss_sweep(40, sync=False)
        b.ss_sweep(40, sync=False)
    a.sync()
    b.sync()
    a2 = make_engine(chains, 1, y, X, obs, prior, ss, sig_up, g0, False)
    a2.ss_sweep(120)
    same_chains(a, a2, [0, 500, 1023], "engine a")
    a2.close()
    b2 = make_engine(chains, 2, y, X, obs, prior, ss, sig_up, g0, False)
    b2.ss_sweep(120)
    same_chains(b, b2, [0, 500, 1023], "engine b")
    for e in (a, b, b2):
        e.close()
