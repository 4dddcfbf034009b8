"""A draw that needs more uniforms than its slot of a substream holds goes on in the slot's
SPILL stream (device_rng.h; the oracle's bo_rng_slot): the probit / logit / Polya-Gamma /
Poisson imputers (4096 / 256 / 4096 / 256 positions an observation; until round 5 the state
draw's Kinderman-Ramage normals too, 256 a normal -- they are Box-Muller pairs now and read two
uniforms for two draws, so the state-draw tests below only check that the switch leaves them
alone).  At those strides that is an event of probability < 1e-40, so the
path is FORCED here: ba_set_slot_limit / bo_set_slot_limit let a slot serve only a few
numbers, every slow normal and nearly every imputation then reads its spill stream, and
the parity tests of each family run once more under that switch -- the device against the
oracle, as exactly as without it.  (Rounds 1-3 stopped a chain that outran a slot.)"""
import numpy as np
import pytest

import test_logit_gpu as tl
import test_poisson_gpu as tpo
import test_polya_gamma as tpg
import test_probit_gpu as tpr
import test_state_space_gpu as tss
import test_structural_general_gpu as tsg
import test_structural_gpu as tst

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[2, 6])
def small_slots(request, oracle, monkeypatch):
    import boom_amd
    limit = request.param
    orig = boom_amd.Engine.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.set_slot_limit(limit)

    monkeypatch.setattr(boom_amd.Engine, "__init__", init)
    oracle.set_slot_limit(limit)
    yield limit
    oracle.set_slot_limit(0)


def test_the_switch_changes_the_draws(oracle):
    """(so that the tests below do test something) the same probit chain with and without it --
    and NOT the state draw: since round 6 its normals are Box-Muller pairs (two uniforms for two
    draws, stream_normals.h), which cannot outrun a slot"""
    from cases import bsts_priors, probit_data, probit_slab, state_space_data
    X, y, nt, _ = probit_data(300, 10, 3, seed=4, max_trials=3)
    slab, pi = probit_slab(X, nt, 3)
    g0 = np.zeros(10, np.uint8)
    g0[0] = 1
    import boom_amd
    eng = []
    for limit in (0, 2):
        e = boom_amd.Engine(2, seed=3)
        e.probit_set_data(X, y, nt, 3)
        e.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
        e.set_spike(pi)
        e.set_state(g0)
        e.set_slot_limit(limit)
        e.probit_sweep(3)
        eng.append(e)
    assert not np.array_equal(eng[0].get_states()[1], eng[1].get_states()[1])
    X, y, _, obs = state_space_data(150, 5, 2, seed=2)
    prior, ss, sig_up = bsts_priors(X, y, 2)
    g0 = np.zeros(5, np.uint8)
    a = tss.make_engine(2, 3, y, X, obs, prior, ss, sig_up, g0)
    b = tss.make_engine(2, 3, y, X, obs, prior, ss, sig_up, g0)
    b.set_slot_limit(2)
    a.ss_sweep(2)
    b.ss_sweep(2)
    assert np.array_equal(a.ss_get_state(0)["state"], b.ss_get_state(0)["state"])


def test_state_draw_local_level(oracle, small_slots):
    tss.test_state_space_every_sweep(oracle, 0.05)
    tss.test_state_space_zero_variances(oracle, 120, "known_initial_state")    # lane-major kernel, prepared normals
    tss.test_state_space_zero_variances(oracle, 2100, "level_fixed_at_zero")   # natural layout


def test_state_draw_structural(oracle, small_slots):
    tst.test_structural_sweeps_match_oracle(oracle, 2, 7, 90, 0.05)
    tsg.test_general_shapes_match_oracle(oracle, [("seasonal", 4, 3, 2), ("trend",), ("ar", 2)], 80, 0.05)


def test_probit_imputer(oracle, small_slots):
    tpr.test_probit_sweeps_match_oracle(oracle, 300, 10, 3, 8, 3)


def test_logit_imputers(oracle, small_slots):
    tl.test_logit_sweeps_match_oracle(oracle, 300, 10, 3, 4, -1)
    tl.test_logit_large_sample_imputation_matches_oracle(oracle, 250, 8, 3, 60, 5)
    tpg.test_device_pg_sweeps_match_the_cpu_twin(oracle, 300, 10, 3, 4, 5)


def test_poisson_imputer(oracle, small_slots):
    tpo.test_poisson_sweeps_match_oracle(oracle, "poisson_small_counts")
    tpo.test_poisson_sweeps_match_oracle(oracle, "poisson_exposure")
