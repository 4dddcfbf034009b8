"""Polya-Gamma augmentation for the logit spike-and-slab sampler (BASELINE config 5 as
worded).  PARITY UNPINNED BY CONSTRUCTION: the reference has no Polya-Gamma sampler
(SURVEY fact 3), so there is nothing to compare draws with.  What can be checked:

* the PG(n, z) draws themselves against the distribution's exact moments (CPU);
* the posterior the PG chain samples against the posterior of the reference's own
  auxiliary-mixture sampler (the oracle's restatement of it IS pinned on the compiled
  reference: tests/golden/logit_*.npz), on the logit goldens' data, within 3 standard
  errors (CPU);
* the device kernel against its CPU twin draw for draw, and many device chains against
  the long auxiliary-mixture run (GPU).
"""
import ctypes as C

import numpy as np
import pytest

from cases import logit_data, probit_slab
from test_oracle_golden import load


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def test_pg_draws_have_the_exact_moments(oracle):
    """E PG(n, z) = n tanh(z/2) / (2 z), Var = n (sinh z - z) / (4 z^3 cosh^2(z/2))"""
    L = oracle.lib
    L.bo_test_rpg.restype = C.c_double
    L.bo_test_rpg.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_long, C.POINTER(C.c_int)]
    N = 60000
    for n, z, clt in [(1, 0.0, 100), (1, 0.7, 100), (1, -3.0, 100), (4, 1.5, 100), (1, 18.0, 100),
                      (40, 0.9, 5)]:   # (the last: the normal draw beyond clt trials)
        r = oracle.rng_philox(11, 0, 10, 0)
        st = C.c_int(0)
        x = np.array([L.bo_test_rpg(C.byref(r), n, z, clt, C.byref(st)) for _ in range(N)])
        assert st.value == 0 and np.all(x > 0)
        az = abs(z)
        if az == 0:
            m, v = n / 4.0, n / 24.0
        else:
            m = n * np.tanh(az / 2) / (2 * az)
            v = n * (np.sinh(az) - az) / (4 * az ** 3 * np.cosh(az / 2) ** 2)
        assert abs(x.mean() - m) < 4 * np.sqrt(v / N), (n, z)
        # the variance of a sample variance: roughly (kurtosis - 1) v^2 / N; PG is right-skewed
        assert abs(x.var() - v) < 0.05 * v, (n, z)


def _chain_means(run, nchains, nsw, burn):
    g, b = [], []
    for c in range(nchains):
        o = run(c)
        assert o["status"] == 0
        g.append(o["gamma"][burn:].mean(0))
        b.append(o["beta"][burn:].mean(0))
    g, b = np.array(g), np.array(b)
    return (g.mean(0), g.std(0, ddof=1) / np.sqrt(nchains)), (b.mean(0), b.std(0, ddof=1) / np.sqrt(nchains))


@pytest.mark.parametrize("name", ["logit_bernoulli", "logit_binomial4"])
def test_pg_posterior_matches_the_auxiliary_mixture_sampler(oracle, name):
    """the two augmentations target the same posterior (the mixture approximates the
    logistic density to ~1e-4): inclusion probabilities and coefficient means of 24
    independent chains each, on the data of the reference goldens, within 3 standard
    errors of the difference"""
    g = load(name)
    X, y, nt = g["X"], g["y"], g["ntrials"]
    slab, pi = dict(mu=g["mu"], prec=g["prec"]), g["pi"]
    p = X.shape[1]
    g0 = g["init_gamma"]
    nch, nsw, burn = 24, 1500, 200

    def run(imputer):
        return lambda c: oracle.logit_run(X, y, nt, slab, pi, ("philox", 101 + imputer, c), g0,
                                          np.zeros(p), nsw, imputer=imputer)
    (ga, sga), (ba, sba) = _chain_means(run(0), nch, nsw, burn)
    (gp, sgp), (bp, sbp) = _chain_means(run(1), nch, nsw, burn)
    zg = np.abs(ga - gp) / np.sqrt(sga ** 2 + sgp ** 2 + 1e-10)
    zb = np.abs(ba - bp) / np.sqrt(sba ** 2 + sbp ** 2 + 1e-10)
    assert np.all(zg < 3.0), (ga, gp, zg)
    assert np.all(zb < 3.0), (ba, bp, zb)
    # ... and they found the same model: the signals in, the rest mostly out
    assert np.array_equal(ga > 0.5, gp > 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("n,p,nsig,max_trials,clt", [(300, 10, 3, 1, 5), (300, 10, 3, 4, 5),
                                                      (200, 8, 3, 30, 6)])
def test_device_pg_sweeps_match_the_cpu_twin(oracle, n, p, nsig, max_trials, clt):
    """device (logit_pg_impute_kernel) against the oracle's restatement of the same
    published algorithm on the same Philox substreams: inclusion indicators bit-exact,
    coefficients within 1e-8 -- the parity bar of every other path, here between the two
    implementations of a sampler the reference does not have"""
    import boom_amd
    X, y, nt, _ = logit_data(n, p, nsig, seed=5 + max_trials + p, max_trials=max_trials)
    slab, pi = probit_slab(X, nt, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 6, 31, 25
    eng = boom_amd.Engine(chains, seed=seed)
    eng.logit_set_data(X, y, nt, clt)
    eng.logit_set_imputer(1)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    eng.set_state(g0)
    check = [0, chains - 1]
    ora = {c: oracle.logit_run(X, y, nt, slab, pi, ("philox", seed, c), g0, np.zeros(p), nsw,
                               clt_threshold=clt, imputer=1) for c in check}
    for s in range(nsw):
        eng.logit_sweep(1)
        gam, beta, _ = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert relerr(beta[c], o["beta"][s]) < 1e-8, (c, s)


@pytest.mark.gpu
def test_device_pg_posterior_matches_the_reference_pinned_sampler(oracle):
    """256 device chains with the Polya-Gamma imputer against a long run of the oracle's
    auxiliary-mixture sampler in MT mode (the mode pinned draw for draw on the compiled
    reference): posterior inclusion probabilities and coefficient means within 3 standard
    errors"""
    import boom_amd
    n, p, nsig = 250, 7, 3
    X, y, nt, _ = logit_data(n, p, nsig, seed=21, max_trials=2)
    slab, pi = probit_slab(X, nt, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    nsw, burn = 6000, 300
    o = oracle.logit_run(X, y, nt, slab, pi, ("mt", 77), g0, np.zeros(p), nsw)
    assert o["status"] == 0
    chains, dburn, rounds = 256, 100, 15
    eng = boom_amd.Engine(chains, seed=3)
    eng.logit_set_data(X, y, nt, 5)
    eng.logit_set_imputer(1)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    eng.set_state(g0)
    eng.logit_sweep(dburn)
    inc, b = [], []
    for _ in range(rounds):
        eng.logit_sweep(8)
        gam, beta, _ = eng.get_states()
        inc.append(gam.astype(float))
        b.append(beta)

    def batch_se(x, nb=20):
        m = np.array([v.mean(0) for v in np.array_split(np.asarray(x, float), nb)])
        return m.std(0, ddof=1) / np.sqrt(nb)

    for dev, ora in ((np.array(inc), o["gamma"][burn:].astype(float)), (np.array(b), o["beta"][burn:])):
        per_chain = dev.mean(0)
        m_d, se_d = per_chain.mean(0), per_chain.std(0, ddof=1) / np.sqrt(chains)
        m_o, se_o = ora.mean(0), batch_se(ora)
        z = np.abs(m_d - m_o) / np.sqrt(se_d ** 2 + se_o ** 2 + 1e-12)
        assert np.all(z < 3.0), (m_d, m_o, z)
