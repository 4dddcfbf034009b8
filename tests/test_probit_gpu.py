"""BinomialProbitSpikeSlabSampler on the device (the probit
member): truncated-normal data augmentation, X'z by one MFMA GEMM, and the
SpikeSlabSampler mode of the sweep kernel, against the CPU oracle (pinned on the
reference: tests/golden/probit_*.npz), through the C-ABI.

The sampler carries continuous latent data, so rounding differences between two
implementations grow from sweep to sweep (~3x per sweep): free-running
comparisons stop after 10 sweeps; longer runs are compared one sweep at a time
from the oracle's state (the per-sweep map is what has to agree).
"""
import numpy as np
import pytest

from cases import probit_data, probit_slab

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def make_engine(chains, seed, X, y, nt, slab, pi, g0, clt=5, max_flips=-1):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed)
    eng.probit_set_data(X, y, nt, clt)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False, max_flips=max_flips)
    eng.set_spike(pi)
    eng.set_state(g0)
    return eng


@pytest.mark.parametrize("n,p,nsig,max_trials,clt", [(300, 10, 3, 1, 5), (300, 10, 3, 8, 3),
                                                      (777, 24, 5, 1, 5), (250, 12, 4, 12, 5)])
def test_probit_sweeps_match_oracle(oracle, n, p, nsig, max_trials, clt):
    X, y, nt, _ = probit_data(n, p, nsig, seed=5 + max_trials, max_trials=max_trials)
    slab, pi = probit_slab(X, nt, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 6, 17, 10
    eng = make_engine(chains, seed, X, y, nt, slab, pi, g0, clt)
    check = [0, chains - 1]
    ora = {c: oracle.probit_run(X, y, nt, slab, pi, ("philox", seed, c), g0, np.zeros(p), nsw,
                                clt_threshold=clt) for c in check}
    for s in range(nsw):
        eng.probit_sweep(1)
        gam, beta, _ = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert relerr(beta[c], o["beta"][s]) < RTOL, (c, s)


def test_probit_recovers_the_signals():
    """a longer run: the three signals are in, the noise variables mostly out, the
    coefficients near the truth (a size-independent property)"""
    n, p = 4000, 32
    X, y, nt, btrue = probit_data(n, p, 4, seed=9)
    slab, pi = probit_slab(X, nt, 4)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng = make_engine(64, 3, X, y, nt, slab, pi, g0)
    eng.probit_sweep(150)
    eng.reset_summaries()
    eng.probit_sweep(100)
    gam, beta, _ = eng.get_states()
    sm = eng.get_summaries()
    inc = sm["inclusion_count"] / sm["sweeps"]
    assert inc[:4].min() > 0.95 and inc[4:].max() < 0.3
    assert np.max(np.abs(beta[:, :4].mean(axis=0) - btrue[:4])) < 0.15
