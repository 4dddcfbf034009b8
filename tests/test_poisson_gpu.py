"""PoissonRegressionSpikeSlabSampler on the device (the Poisson member):
the auxiliary-mixture imputation (last event time by Cheng's beta sampler, the event past
the interval, two unmixing draws against the reference table's mixtures), X'Wz by one MFMA
GEMM, every chain's X'WX a vector at a time, SpikeSlabSampler's inclusion / coefficient
draws -- against the CPU oracle (pinned on the compiled reference:
tests/golden/poisson_*.npz), through the C-ABI.  The mixtures are fixture data read from
the goldens (generated from the reference's own table, tests/golden/make_golden_poisson.py).

Bar: inclusion indicators bit-exact, coefficients within 1e-8 relative.
"""
import numpy as np
import pytest

from test_oracle_golden import _golden_mix, load

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


@pytest.mark.parametrize("name", ["poisson_small_counts", "poisson_exposure", "poisson_large_counts",
                                  "poisson_p24_maxflips"])
def test_poisson_sweeps_match_oracle(oracle, name):
    import boom_amd
    g = load(name)
    X, y, ex = g["X"], g["y"], g["exposure"]
    slab, pi, mix = dict(mu=g["mu"], prec=g["prec"]), g["pi"], _golden_mix(g)
    p = X.shape[1]
    g0 = g["init_gamma"]
    mf = int(g["max_flips"])
    chains, seed, nsw = 6, 19, 25
    eng = boom_amd.Engine(chains, seed=seed)
    eng.poisson_set_data(X, y, ex, mix)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False, max_flips=mf)
    eng.set_spike(pi)
    eng.set_state(g0)
    check = [0, chains - 1]
    ora = {c: oracle.poisson_run(X, y, ex, slab, pi, mix, ("philox", seed, c), g0, np.zeros(p), nsw,
                                 max_flips=mf) for c in check}
    for s in range(nsw):
        eng.poisson_sweep(1)
        gam, beta, _ = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert relerr(beta[c], o["beta"][s]) < RTOL, (c, s)
    eng2 = boom_amd.Engine(chains, seed=seed)
    eng2.poisson_set_data(X, y, ex, mix)
    eng2.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False, max_flips=mf)
    eng2.set_spike(pi)
    eng2.set_state(g0)
    eng2.poisson_sweep(nsw)
    a, b = eng.get_states(), eng2.get_states()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_poisson_recovers_the_signals_and_rejects_bad_arguments():
    import boom_amd
    g = load("poisson_exposure")
    X, y, ex, mix = g["X"], g["y"], g["exposure"], _golden_mix(g)
    eng = boom_amd.Engine(64, seed=5)
    with pytest.raises(boom_amd.BoomAmdError):
        eng.poisson_sweep(1)                              # no data
    bad = dict(mix)
    keep = mix["counts"] != int(y[y > 0][0])
    off = np.concatenate([[0], np.cumsum(mix["ncomp"])])
    sel = np.concatenate([np.arange(off[i], off[i + 1]) for i in range(len(keep)) if keep[i]])
    bad.update(counts=mix["counts"][keep], ncomp=mix["ncomp"][keep], mu=mix["mu"][sel],
               sigma=mix["sigma"][sel], weight=mix["weight"][sel])
    with pytest.raises(boom_amd.BoomAmdError) as ei:
        eng.poisson_set_data(X, y, ex, bad)               # a count of the data has no mixture
    assert "no mixture" in str(ei.value)
    eng.poisson_set_data(X, y, ex, mix)
    eng.sss_set_slab(g["mu"], g["prec"], scales_with_sigsq=False)
    eng.set_spike(g["pi"])
    eng.set_state(g["init_gamma"])
    with pytest.raises(boom_amd.BoomAmdError):
        eng.logit_sweep(1)                                # Poisson data: its own sweep
    eng.poisson_sweep(150)
    eng.reset_summaries()
    eng.poisson_sweep(100)
    sm = eng.get_summaries()
    inc = sm["inclusion_count"] / sm["sweeps"]
    assert inc[:3].min() > 0.95 and inc[3:].max() < 0.3
