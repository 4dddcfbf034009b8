"""BinomialLogitSpikeSlabSampler on the device (the logit
member; the sampler behind BASELINE config 5 with the reference's own
auxiliary-mixture imputer): per-trial truncated logistic + mixture component,
X'Wz by one MFMA GEMM, every chain's X'WX by a batched weighted MFMA syrk, the
sampler's inclusion / coefficient draws -- against the CPU oracle (pinned on the
reference: tests/golden/logit_*.npz), through the C-ABI.

Bar: inclusion indicators bit-exact, coefficients within 1e-8 relative.
"""
import numpy as np
import pytest

from cases import logit_data, probit_slab

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def make_engine(chains, seed, X, y, nt, slab, pi, g0, clt=5, max_flips=-1, **kw):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed, **kw)
    eng.logit_set_data(X, y, nt, clt)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False, max_flips=max_flips)
    eng.set_spike(pi)
    eng.set_state(g0)
    return eng


@pytest.mark.parametrize("n,p,nsig,max_trials,max_flips",
                         [(300, 10, 3, 1, -1), (300, 10, 3, 4, -1), (777, 24, 5, 1, 9),
                          (500, 70, 6, 1, -1), (64, 5, 2, 3, -1)])
def test_logit_sweeps_match_oracle(oracle, n, p, nsig, max_trials, max_flips):
    X, y, nt, _ = logit_data(n, p, nsig, seed=5 + max_trials + p, max_trials=max_trials)
    slab, pi = probit_slab(X, nt, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 6, 17, 25
    eng = make_engine(chains, seed, X, y, nt, slab, pi, g0, max_flips=max_flips)
    check = [0, chains - 1]
    ora = {c: oracle.logit_run(X, y, nt, slab, pi, ("philox", seed, c), g0, np.zeros(p), nsw,
                               max_flips=max_flips) for c in check}
    for s in range(nsw):
        eng.logit_sweep(1)
        gam, beta, _ = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert relerr(beta[c], o["beta"][s]) < RTOL, (c, s)
    # several sweeps in one call are the same draws
    eng2 = make_engine(chains, seed, X, y, nt, slab, pi, g0, max_flips=max_flips)
    eng2.logit_sweep(nsw)
    a, b = eng.get_states(), eng2.get_states()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_logit_capacity_escalation_and_recovery(oracle):
    """20 signals from a one-variable start: chains outgrow the 16-variable launch
    capacity inside a sweep and replay it on the same latent data"""
    n, p, nsig = 1500, 40, 8
    rng = np.random.Generator(np.random.PCG64(2))
    X = rng.standard_normal((n, p))
    X[:, 0] = 1.0
    btrue = np.zeros(p)
    btrue[:20] = rng.choice([-1.0, 1.0], 20) * 1.2
    y = rng.binomial(1, 1 / (1 + np.exp(-(X @ btrue)))).astype(float)
    nt = np.ones(n)
    slab, pi = probit_slab(X, nt, 20)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 4, 5, 12
    import boom_amd
    eng = make_engine(chains, seed, X, y, nt, slab, pi, g0)
    eng.set_tuning(kcap_start=16)
    ora = {c: oracle.logit_run(X, y, nt, slab, pi, ("philox", seed, c), g0, np.zeros(p), nsw)
           for c in (0, 3)}
    eng.logit_sweep(nsw)
    gam, beta, _ = eng.get_states()
    for c in (0, 3):
        assert np.array_equal(gam[c], ora[c]["gamma"][-1]), c
        assert relerr(beta[c], ora[c]["beta"][-1]) < RTOL, c
    assert gam.sum(axis=1).min() > 16


def test_logit_models_of_more_than_64_variables(oracle):
    """72 signals: the chains move on to the kernel whose factors live in HBM, with
    their own V matrices and the logit sampler's shuffle"""
    n, p, nsig = 2500, 96, 72
    rng = np.random.Generator(np.random.PCG64(4))
    X = rng.standard_normal((n, p))
    X[:, 0] = 1.0
    btrue = np.zeros(p)
    btrue[:nsig] = rng.choice([-1.0, 1.0], nsig) * 1.5
    y = rng.binomial(1, 1 / (1 + np.exp(-(X @ btrue)))).astype(float)
    nt = np.ones(n)
    slab, pi = probit_slab(X, nt, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 3, 11, 14
    eng = make_engine(chains, seed, X, y, nt, slab, pi, g0)
    ora = {c: oracle.logit_run(X, y, nt, slab, pi, ("philox", seed, c), g0, np.zeros(p), nsw)
           for c in (0, 2)}
    for s in range(nsw):
        eng.logit_sweep(1)
        gam, beta, _ = eng.get_states()
        for c in (0, 2):
            assert np.array_equal(gam[c], ora[c]["gamma"][s]), (c, s)
            assert relerr(beta[c], ora[c]["beta"][s]) < RTOL, (c, s)
    assert max(o["gamma"].sum(axis=1).max() for o in ora.values()) > 64


def test_logit_rejects_bad_arguments():
    """the reference's own argument checks (BinomialLogitDataImputer.cpp:42-60) and the
    bound on clt_threshold that the observation substreams impose"""
    import boom_amd
    X, y, nt, _ = logit_data(50, 4, 2, seed=1, max_trials=9)
    eng = boom_amd.Engine(2, seed=1)
    with pytest.raises(boom_amd.BoomAmdError) as ei:
        eng.logit_set_data(X, y, nt, 65)
    assert "clt_threshold" in str(ei.value)
    bad = y.copy()
    bad[3] = nt[3] + 1
    with pytest.raises(boom_amd.BoomAmdError) as ei:
        eng.logit_set_data(X, bad, nt, 5)
    assert "must not exceed the number of trials" in str(ei.value)
    eng.logit_set_data(X, y, nt, 5)      # trial counts above the threshold are served now


@pytest.mark.parametrize("imputer", [0, 1])
def test_config5_shape_per_gpu(imputer):
    """BASELINE config 5 at its per-GPU size (n = 5e4, p = 1024, 4096 chains over 8 GPUs
    = 512 per rank), too large for an oracle run: the eight signals are found by every
    chain, and a shard of four chains repeats the whole job's first four bit for bit --
    which it can only do if no value depends on what else shares a launch (the request
    GEMM's fixed row chunks, the replay of parked chains).  imputer 0: the reference's
    mixture-of-normals imputation, 1: the Polya-Gamma imputer (logit_pg_impute_kernel)."""
    import boom_amd
    n, p, nsig = 50000, 1024, 8
    X, y, nt, _ = logit_data(n, p, nsig, seed=8675309)
    slab, pi = probit_slab(X, nt, nsig)
    out = []
    for chains in (512, 4):
        eng = boom_amd.Engine(chains, seed=4)
        eng.logit_set_data(X, y, nt, 5)
        eng.logit_set_imputer(imputer)
        eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
        eng.set_spike(pi)
        g0 = np.zeros(p, np.uint8)
        g0[0] = 1
        eng.set_state(g0)
        eng.logit_sweep(12)
        out.append(eng.get_states()[:2])
        del eng
    (gam, beta), (gam4, beta4) = out
    assert gam[:, :nsig].all()
    assert gam.sum(1).max() < 40
    assert np.array_equal(gam[:4], gam4) and np.array_equal(beta[:4], beta4)


def test_plain_sweep_is_refused_while_binomial_data_are_set():
    """SpikeSlabSampler's sweep without the imputation is not a draw of the binomial
    samplers (and the logit sampler's V holds only the vectors its last sweep asked
    for): refused until regression data are installed again."""
    import boom_amd
    from boom_amd.capi import BoomAmdError
    X, y, nt, _ = logit_data(200, 6, 2, seed=3)
    slab, pi = probit_slab(X, nt, 2)
    eng = make_engine(4, 5, X, y, nt, slab, pi, np.zeros(6, np.uint8))
    eng.logit_sweep(2)
    with pytest.raises(BoomAmdError, match="binomial data are set"):
        eng.sss_sweep(1)
    eng.build_suf_from_xy(X, y)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    eng.set_state(np.zeros(6, np.uint8), sigsq=1.0)
    eng.sss_sweep(3)


@pytest.mark.parametrize("kind", ["logit", "probit"])
def test_posterior_agrees_with_the_sequential_stream_sampler(oracle, kind):
    """The imputers read one substream position per observation on the device (and in the
    oracle's Philox mode); the oracle's MT mode -- pinned draw for draw on the compiled
    reference -- reads ONE stream over the observations.  Same transforms on independent
    uniforms: posterior means of many device chains against a long MT run, within 5
    standard errors."""
    import boom_amd
    from cases import probit_data
    n, p, nsig = 250, 7, 3
    X, y, nt, _ = (logit_data if kind == "logit" else probit_data)(n, p, nsig, seed=21, max_trials=2)
    slab, pi = probit_slab(X, nt, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    nsw, burn = 5000, 300
    run = oracle.logit_run if kind == "logit" else oracle.probit_run
    o = run(X, y, nt, slab, pi, ("mt", 77), g0, np.zeros(p), nsw)
    assert o["status"] == 0
    chains, dburn, rounds = 256, 100, 15
    eng = boom_amd.Engine(chains, seed=3)
    (eng.logit_set_data if kind == "logit" else eng.probit_set_data)(X, y, nt, 5)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    eng.set_state(g0)
    sweep = eng.logit_sweep if kind == "logit" else eng.probit_sweep
    sweep(dburn)
    inc, b = [], []
    for _ in range(rounds):
        sweep(8)
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
        assert np.all(z < 5.0), (kind, m_d, m_o, z)


@pytest.mark.parametrize("n,p,nsig,max_trials,clt", [(250, 8, 3, 60, 5), (120, 6, 2, 200, 10),
                                                      (400, 12, 4, 30, 64)])
def test_logit_large_sample_imputation_matches_oracle(oracle, n, p, nsig, max_trials, clt):
    """Observations with more than clt_threshold trials: BinomialLogitCltDataImputer::
    impute_large_sample (BinomialLogitDataImputer.cpp:155-211) on the device -- two
    multinomial draws over the nine mixture components (BTPE / inversion binomials) and
    one normal draw per observation.  The oracle's restatement is pinned on the compiled
    reference by tests/golden/logit_binomial{60,200}_large_sample.npz; the last case
    keeps every observation on the per-trial branch up to 30 trials (clt_threshold 64)."""
    X, y, nt, _ = logit_data(n, p, nsig, seed=5 + max_trials + p, max_trials=max_trials)
    slab, pi = probit_slab(X, nt, nsig)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    chains, seed, nsw = 5, 23, 20
    eng = make_engine(chains, seed, X, y, nt, slab, pi, g0, clt=clt)
    check = [0, chains - 1]
    ora = {c: oracle.logit_run(X, y, nt, slab, pi, ("philox", seed, c), g0, np.zeros(p), nsw,
                               clt_threshold=clt) for c in check}
    for s in range(nsw):
        eng.logit_sweep(1)
        gam, beta, _ = eng.get_states()
        for c in check:
            o = ora[c]
            assert o["status"] == 0
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert relerr(beta[c], o["beta"][s]) < RTOL, (c, s)
