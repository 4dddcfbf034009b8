"""Several devices behind the C-ABI (ba_group_*: one process, one engine per device-list
entry, librccl for the two collectives) -- on the one GPU a test box has: a device list
that names device 0 twice.  That path needs no collective (the blocks are summed on the
device), but everything else is what an 8-GPU group does: global chain ids, the
row-sharded data build, per-engine streams, the gathered summary blocks and their
aggregate.  VERDICT r2 item 6.

Checked: the group's two engines of 512 chains ARE chains 0..1023 of a single engine of
1024 (same states bit for bit, hence the same chains as the oracle's); the row-sharded
sufficient statistics equal the single-shot build to rounding and are bitwise the same
on both engines; the whole-job summaries equal the single engine's (counts exactly, sums
to rounding); ba_group_locate; argument errors.
"""
import numpy as np
import pytest

from cases import regression_data, spike_slab_prior
from oracle_lib import ssvs_options

pytestmark = pytest.mark.gpu


def _suf(e):
    s = e.get_suf()
    return dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"],
                xsum=s["xbar"] * s["n"])


def _device_count():
    # (counting devices does not initialise the GPU runtime in this process)
    import torch
    return torch.cuda.device_count()


def test_group_of_two_engines_is_one_job(oracle):
    _group_is_one_job(oracle, [0, 0])


def test_group_over_two_devices_runs_the_rccl_collectives(oracle):
    """VERDICT r5 task 7: the SAME checks over the device list [0, 1] -- there the row-sharded
    build's ncclAllReduce and the summaries' ncclAllGather (group.hip, librccl by dlopen:
    ncclCommInitAll over the list) really run.  Skipped, with the reason, on a one-GPU box: the
    first lease with two devices executes that code before any scaling run does."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs: %d visible (the RCCL path of ba_group_* runs from two devices on)" % _device_count())
    _group_is_one_job(oracle, [0, 1])


def _group_is_one_job(oracle, devices):
    import boom_amd
    n, p, nsig, per, seed, nsw = 3001, 48, 6, 512, 77, 120
    X, y, _ = regression_data(n, p, nsig, seed=9)
    grp = boom_amd.Group(devices, per, seed=seed)
    assert grp.size == 2
    assert grp.locate(0) == (0, 0) and grp.locate(per) == (1, 0) and grp.locate(2 * per - 1) == (1, per - 1)
    with pytest.raises(boom_amd.BoomAmdError):
        grp.locate(2 * per)
    grp.build_suf_from_xy(X, y)
    s0, s1 = grp.engines[0].get_suf(), grp.engines[1].get_suf()
    for k in ("xtx", "xty", "xbar"):
        assert np.array_equal(s0[k], s1[k]), k            # every engine holds the same statistics
    assert s0["yty"] == s1["yty"] and s0["n"] == n
    one = boom_amd.Engine(2 * per, seed=seed)
    one.build_suf_from_xy(X, y)
    ref = one.get_suf()
    assert np.max(np.abs(s0["xtx"] - ref["xtx"])) < 1e-12 * np.abs(ref["xtx"]).max()
    assert np.max(np.abs(s0["xty"] - ref["xty"])) < 1e-12 * np.abs(ref["xty"]).max()
    # the single engine on the group's statistics, so that chains can be compared bit for bit
    one.upload_suf(s0["xtx"], s0["xty"], s0["yty"], s0["n"], s0["ybar"], s0["xbar"])
    suf = _suf(grp.engines[0])
    prior = spike_slab_prior(suf, nsig)
    grp.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    one.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    grp.set_state(g0)
    one.set_state(g0)
    grp.sweep(nsw)
    one.sweep(nsw)
    go, bo, so = one.get_states()
    for i, e in enumerate(grp.engines):
        g, b, s = e.get_states()
        assert np.array_equal(g, go[i * per:(i + 1) * per])
        assert np.array_equal(b, bo[i * per:(i + 1) * per])
        assert np.array_equal(s, so[i * per:(i + 1) * per])
    # ... and they are the reference's chains: global ids key the streams
    for c in (0, per, 2 * per - 1):
        o = oracle.ssvs_run(suf, prior, ssvs_options(), ("philox", seed, c), g0, nsw)
        ei, lc = grp.locate(c)
        g, b, s = grp.engines[ei].get_state(lc)
        assert np.array_equal(g, o["gamma"][-1])
        assert np.max(np.abs(b - o["beta"][-1]) / np.maximum(np.abs(o["beta"][-1]), 1e-3)) < 1e-8
    sg, s1 = grp.get_summaries(), one.get_summaries()
    assert sg["sweeps"] == s1["sweeps"] == 2 * per * nsw
    assert np.array_equal(sg["inclusion_count"], s1["inclusion_count"])
    assert sg["accepts"] == s1["accepts"] and sg["proposals"] == s1["proposals"]
    assert sg["min_margin"] == s1["min_margin"] and sg["k_sum"] == s1["k_sum"]
    assert np.allclose(sg["beta_sum"], s1["beta_sum"], rtol=1e-12, atol=1e-9)
    assert sg["blocks"].shape == (2, 3 * p + 16)
    assert sg["blocks"][:, 3 * p].tolist() == [per * nsw, per * nsw]
    # the per-device blocks add up to the aggregate
    assert np.array_equal(sg["blocks"][:, :p].sum(0), sg["inclusion_count"])
    grp.reset_summaries()
    grp.sweep(3)
    assert grp.get_summaries()["sweeps"] == 2 * per * 3
    grp.close()
    one.close()


def test_group_argument_errors():
    import boom_amd
    with pytest.raises(boom_amd.BoomAmdError):
        boom_amd.Group([0, 0], 0)
    with pytest.raises(boom_amd.BoomAmdError):
        boom_amd.Group([99], 4)          # no such device
    g = boom_amd.Group([0], 8)
    with pytest.raises(boom_amd.BoomAmdError):
        g.get_summaries()                 # no data yet
    g.close()


def test_librccl_is_loadable_with_every_symbol_the_group_uses():
    """the multi-device path itself needs more than one GPU; what can be checked on one is
    that the collective library resolves (dlopen + the seven entry points)"""
    import boom_amd
    lib = boom_amd.load_library()
    assert lib.ba_group_rccl_available() == 1, lib.ba_group_last_error().decode()


def test_group_runs_the_bsts_and_logit_samplers(oracle):
    """the samplers whose data are replicated -- bsts (any state list, with the look-ahead)
    and logit -- behind one group handle: ba_group_call sets every engine up, the group's
    sweeps enqueue on every device before any is waited for, and the group's two engines of
    C chains ARE chains 0 .. 2C - 1 of one engine of 2C chains, bit for bit"""
    import boom_amd
    from cases import bsts_priors, general_data, general_spec, logit_data, probit_slab
    # ---- bsts: a seasonal block of duration 3, a trend, served from the look-ahead
    T, p, per, seed = 80, 5, 6, 41
    X, y, _, obs = general_data(T, p, 2, [(4, 3)], seed=3, missing_frac=0.03)
    prior, _, sig_up = bsts_priors(X, y, 2)
    blocks = general_spec(y, [("seasonal", 4, 3), ("trend",)])
    g0 = np.zeros(p, np.uint8)

    def setup(e):
        e.ss_set_data(y, X, obs)
        e.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                     sigma_upper_limit=sig_up)
        e.ss_set_state_models(blocks)
        e.set_state(g0)
    grp = boom_amd.Group([0, 0], per, seed=seed)
    grp.call(setup)
    grp.call(lambda e: e.ss_set_lookahead(4))
    one = boom_amd.Engine(2 * per, seed=seed)
    setup(one)
    for it in range(9):
        grp.ss_draw_next()
        one.ss_sweep(1)
        go, bo, so = one.get_states()
        for i, e in enumerate(grp.engines):
            gg, bg, sg = e.get_states()
            sl = slice(i * per, (i + 1) * per)
            assert np.array_equal(gg, go[sl]) and np.array_equal(bg, bo[sl]) and np.array_equal(sg, so[sl]), (it, i)
    ei, ci = grp.locate(per + 2)
    assert np.array_equal(grp.engines[ei].ss_get_state_draw(ci), one.ss_get_state_draw(per + 2))
    grp.ss_sweep(3)
    one.ss_sweep(3)
    assert np.array_equal(grp.engines[1].get_states()[1], one.get_states()[1][per:])
    # an error inside ba_group_call stops the loop and comes out as the engine's error
    with pytest.raises(boom_amd.BoomAmdError):
        grp.call(lambda e: e.ss_set_lookahead(0))
    grp.close()
    one.close()
    # ---- logit
    n, p, per = 600, 12, 8
    Xl, yl, nt, _ = logit_data(n, p, 3, seed=5)
    slab, pi = probit_slab(Xl, nt, 3)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1

    def setup_logit(e):
        e.logit_set_data(Xl, yl, nt, 5)
        e.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
        e.set_spike(pi)
        e.set_state(g0)
    grp = boom_amd.Group([0, 0], per, seed=seed)
    grp.call(setup_logit)
    one = boom_amd.Engine(2 * per, seed=seed)
    setup_logit(one)
    grp.logit_sweep(12)
    one.logit_sweep(12)
    go, bo, _ = one.get_states()
    for i, e in enumerate(grp.engines):
        gg, bg, _ = e.get_states()
        assert np.array_equal(gg, go[i * per:(i + 1) * per]) and np.array_equal(bg, bo[i * per:(i + 1) * per])
