"""A shard of a job draws what the whole job draws for its chains -- the property
the multi-GPU layout rests on (rank r owns global chains [r C, (r + 1) C), the chain
id keys every stream) -- for the paths added in round 2: structural state space,
probit, logit, adaptive.  One engine of 8 chains against two engines of 4 with
chain_offset 0 and 4."""
import numpy as np
import pytest

from cases import (bsts_priors, logit_data, probit_data, probit_slab, structural_data,
                   structural_spec)

pytestmark = pytest.mark.gpu


def _structural(chains, offset):
    import boom_amd
    T, p = 130, 5
    X, y, _, obs = structural_data(T, p, 2, 7, seed=9, missing_frac=0.04)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, 2, 7)
    eng = boom_amd.Engine(chains, seed=21, chain_offset=offset)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                   sigma_upper_limit=sig_up)
    eng.ss_set_structural(2, 7, spec["var_df"], spec["var_sigma_guess"],
                          spec["var_sigma_upper_limit"], spec["var_initial_sigma"],
                          spec["initial_state_mean"], spec["initial_state_variance"])
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_sweep(8)
    gam, beta, sig = eng.get_states()
    st = np.stack([eng.ss_get_structural(c)["state"] for c in range(chains)])
    return gam, beta, sig, st


def _binomial(kind, chains, offset):
    import boom_amd
    data = probit_data if kind == "probit" else logit_data
    X, y, nt, _ = data(400, 14, 4, seed=12, max_trials=3)
    slab, pi = probit_slab(X, nt, 4)
    eng = boom_amd.Engine(chains, seed=33, chain_offset=offset)
    (eng.probit_set_data if kind == "probit" else eng.logit_set_data)(X, y, nt, 5)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    g0 = np.zeros(14, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    (eng.probit_sweep if kind == "probit" else eng.logit_sweep)(9)
    gam, beta, _ = eng.get_states()
    return gam, beta


def test_structural_shards_equal_the_whole():
    whole = _structural(8, 0)
    lo, hi = _structural(4, 0), _structural(4, 4)
    for w, a, b in zip(whole, lo, hi):
        assert np.array_equal(w[:4], a) and np.array_equal(w[4:], b)


@pytest.mark.parametrize("kind", ["probit", "logit"])
def test_binomial_shards_equal_the_whole(kind):
    whole = _binomial(kind, 8, 0)
    lo, hi = _binomial(kind, 4, 0), _binomial(kind, 4, 4)
    for w, a, b in zip(whole, lo, hi):
        assert np.array_equal(w[:4], a) and np.array_equal(w[4:], b)
