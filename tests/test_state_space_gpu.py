"""GPU parity of the bsts local-level + regression path (Kalman filter +
Durbin-Koopman simulation smoother + per-chain regression sufficient statistics
+ SSVS on them) against the CPU oracle, through the C-ABI.

Bars: inclusion indicators bit-exact; beta, sigma^2_obs, sigma^2_level and the
state draw within RTOL (fp64; the kernel reduces X'e and e'e with a different
summation order than the sequential oracle).
"""
import numpy as np
import pytest

from cases import bsts_priors, state_space_data
from oracle_lib import ssvs_options

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def relerr(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def make_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0):
    import boom_amd
    eng = boom_amd.Engine(chains, seed=seed)
    eng.ss_set_data(y, X, obs)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"],
                           ss["level_sigma_upper_limit"], ss["initial_state_mean"],
                           ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(g0)
    return eng


@pytest.mark.parametrize("missing", [0.0, 0.05])
def test_state_space_every_sweep(oracle, missing):
    T, p, chains, nsw, seed = 200, 8, 6, 40, 31
    X, y, _, obs = state_space_data(T, p, 3, seed=5, missing_frac=missing)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0)
    ora = [oracle.ss_run(y, X, obs, prior, opts, ss, ("philox", seed, c), g0, nsw)
           for c in range(chains)]
    for o in ora:
        assert o["status"] == 0
    for s in range(nsw):
        eng.ss_sweep(1)
        gam, beta, sig = eng.get_states()
        for c in range(chains):
            o = ora[c]
            assert np.array_equal(gam[c], o["gamma"][s]), (c, s)
            assert relerr(beta[c], o["beta"][s]) < RTOL, (c, s)
            assert abs(sig[c] - o["sigsq"][s]) < RTOL * sig[c], (c, s)
            st = eng.ss_get_state(c)
            assert abs(st["level_sigsq"] - o["level_sigsq"][s]) < RTOL * st["level_sigsq"]
            assert np.max(np.abs(st["state"] - o["state"][s])) < 1e-8 * np.abs(o["state"][s]).max()


@pytest.mark.parametrize("T", [120, 2100])
@pytest.mark.parametrize("case", ["known_initial_state", "level_fixed_at_zero"])
def test_state_space_zero_variances(oracle, T, case):
    """A zero standard deviation draws nothing (Bmath/rnorm.cpp:63-64), so the sweep's
    normals are fewer and sit elsewhere on the stream: a known initial state (P0 = 0) and a
    level variance held at zero (upper limit 0), on the lane-major kernel (T <= 2048, normals
    prepared a round ahead in their slots) and on the natural-layout one."""
    p, chains, nsw, seed = 6, 3, 6, 5
    X, y, _, obs = state_space_data(T, p, 2, seed=3, missing_frac=0.05)
    prior, ss, sig_up = bsts_priors(X, y, 2)
    ss = dict(ss)
    if case == "known_initial_state":
        ss["initial_state_variance"] = 0.0
    else:
        ss["level_sigma_upper_limit"] = 0.0
        ss["initial_level_sigma"] = 0.0
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0)
    ora = [oracle.ss_run(y, X, obs, prior, opts, ss, ("philox", seed, c), g0, nsw) for c in range(chains)]
    eng.ss_sweep(2)          # (two rounds in one call: the second one's normals come prepared)
    eng.ss_sweep(nsw - 2)
    gam, beta, sig = eng.get_states()
    for c in range(chains):
        o = ora[c]
        assert o["status"] == 0
        assert np.array_equal(gam[c], o["gamma"][-1]), c
        assert relerr(beta[c], o["beta"][-1]) < RTOL, c
        st = eng.ss_get_state(c)
        assert abs(st["level_sigsq"] - o["level_sigsq"][-1]) <= RTOL * o["level_sigsq"][-1]
        assert np.max(np.abs(st["state"] - o["state"][-1])) < 1e-8 * np.abs(o["state"][-1]).max()


def test_impute_state_sufficient_statistics(oracle):
    """one impute_state with fixed parameters: state draw and the regression /
    level sufficient statistics it leaves behind (a14-a19)."""
    import ctypes as C
    T, p, seed = 200, 8, 77
    X, y, _, obs = state_space_data(T, p, 3, seed=5, missing_frac=0.05)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    beta = np.array([3, 6, 9, 0, 0, 0, 0, 0.])
    gam = (beta != 0).astype(np.uint8)
    ss2 = dict(ss, initial_state_mean=float(y[0]), initial_state_variance=4.0,
               initial_level_sigma=0.5)
    eng = make_engine(2, seed, y, X, obs, prior, ss2, sig_up, gam)
    eng.set_state(gam, beta, 0.04)
    eng.ss_impute_state()
    for c in range(2):
        rng = oracle.rng_philox(seed, chain=c, stream=2)
        o = oracle.ss_impute_state(y, X, obs, beta, gam, 0.04, 0.25, float(y[0]), 4.0, rng)
        st = eng.ss_get_state(c)
        assert np.max(np.abs(st["state"] - o["state"])) < 1e-10
        assert st["level_n"] == o["level_n"]
        assert abs(st["level_sumsq"] - o["level_sumsq"]) < 1e-11 * o["level_sumsq"]
        suf = eng.ss_get_chain_suf(c)
        e = np.where(obs.astype(bool), y - o["state"], 0.0)
        assert relerr(suf["xty"], X.T @ e, 1e-6) < 1e-10
        assert abs(suf["yty"] - e @ e) < 1e-10 * (e @ e)
        assert suf["n"] == obs.sum()


def test_config3_shape_properties():
    """BASELINE config 3 shape (T=2000, p=100, 1024 chains): size-independent
    properties of the draws."""
    T, p, nsig, chains = 2000, 100, 5, 1024
    X, y, btrue, _ = state_space_data(T, p, nsig, seed=8675309)
    prior, ss, sig_up = bsts_priors(X, y, 5)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, 4, y, X, None, prior, ss, sig_up, g0)
    eng.ss_sweep(30)
    gam, beta, sig = eng.get_states()
    assert np.all(np.isfinite(beta)) and np.all(sig > 0)
    assert np.sqrt(sig).max() <= sig_up * (1 + 1e-12)
    assert gam[:, :nsig].mean() > 0.99
    assert np.abs(beta[:, :nsig].mean(axis=0) - btrue[:nsig]).max() < 0.1
    st = eng.ss_get_state(17)
    assert st["level_n"] == T - 1
    assert np.sqrt(st["level_sigsq"]) <= ss["level_sigma_upper_limit"] * (1 + 1e-12)
    # residual sd = observation noise (+ the state draw's posterior spread)
    resid = y - X @ beta[17] - st["state"]
    assert 0.15 < resid.std() < 0.4
    assert 0.1 < np.sqrt(sig).mean() < 0.5   # truth 0.2; 30 sweeps from a cold start


def test_capacity_escalation_in_stream():
    """A chain that outgrows the launch capacity in state-space mode sits out
    the rest of the call and is caught up, one (SSVS, Kalman) pair at a time,
    with a larger capacity: the draws must be those of a run whose capacity was
    large enough from the start."""
    import boom_amd
    T, p, nsig = 300, 48, 26
    X, y, btrue, _ = state_space_data(T, p, nsig, seed=31)
    prior, ss, sig_up = bsts_priors(X, y, nsig)

    def run(hint, start):
        eng = boom_amd.Engine(6, seed=9, max_model_size_hint=hint)
        if start:
            eng.set_tuning(kcap_start=start)
        eng.ss_set_data(y, X, None)
        eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                       prior["sigma_guess"], sigma_upper_limit=sig_up)
        eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"],
                               ss["level_sigma_upper_limit"], ss["initial_state_mean"],
                               ss["initial_state_variance"], ss["initial_level_sigma"])
        eng.set_state(np.zeros(p, np.uint8))
        eng.ss_sweep(25)
        eng.ss_sweep(10)
        gam, beta, sig = eng.get_states()
        st = [eng.ss_get_state(c) for c in range(6)]
        return gam, beta, sig, st

    ref = run(64, 0)
    got = run(0, 16)
    assert ref[0].sum(1).max() > 16     # the models did outgrow the first capacity
    assert np.array_equal(ref[0], got[0])
    assert np.array_equal(ref[1], got[1])
    assert np.array_equal(ref[2], got[2])
    for a, b in zip(ref[3], got[3]):
        assert np.array_equal(a["state"], b["state"])
        assert a["level_sigsq"] == b["level_sigsq"]


def test_forecast_matches_oracle(oracle):
    """f4: simulate_forecast for every chain's current draw -- the next `horizon`
    observations given the chain's beta, sigma^2, level variance and final state,
    normals in the reference's order on the chain's forecast stream"""
    T, p, chains, seed, h = 300, 8, 7, 23, 30
    X, y, _, obs = state_space_data(T, p, 3, seed=5, missing_frac=0.03)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0)
    eng.ss_sweep(25)
    newX = np.random.Generator(np.random.PCG64(8)).standard_normal((h, p))
    f1 = eng.ss_forecast(newX)
    f2 = eng.ss_forecast(newX)          # the streams continue: a second, different draw
    gam, beta, sig = eng.get_states()
    assert f1.shape == (chains, h) and not np.array_equal(f1, f2)
    for c in range(chains):
        st = eng.ss_get_state(c)
        rng = oracle.rng_philox(seed, chain=c, stream=5)
        want1 = oracle.ss_forecast(rng, newX, beta[c], sig[c], st["level_sigsq"], st["state"][-1])
        want2 = oracle.ss_forecast(rng, newX, beta[c], sig[c], st["level_sigsq"], st["state"][-1])
        assert np.max(np.abs(f1[c] - want1)) < 1e-9 * np.abs(want1).max()
        assert np.max(np.abs(f2[c] - want2)) < 1e-9 * np.abs(want2).max()
    # forecasts centre on level + regression: the predictive mean over chains tracks it
    mid = np.array([eng.ss_get_state(c)["state"][-1] for c in range(chains)]).mean()
    assert abs(np.median(f1[:, 0] - newX[0] @ beta.mean(axis=0)) - mid) < 3.0


def test_posterior_agrees_with_the_sequential_stream_sampler(oracle):
    """The device (and the oracle's Philox mode) give every state-stream normal its own
    substream position; the oracle's MT mode -- the one pinned draw for draw on the
    compiled reference -- reads one sequential stream.  Same transforms on independent
    uniforms either way, so the posteriors must agree: long run of the MT oracle against
    many device chains, posterior means within 5 standard errors (batch means)."""
    T, p, seed = 120, 6, 7
    X, y, _, obs = state_space_data(T, p, 2, seed=8, missing_frac=0.03)
    prior, ss, sig_up = bsts_priors(X, y, 2)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    nsw_o, burn_o = 6000, 500
    o = oracle.ss_run(y, X, obs, prior, opts, ss, ("mt", 2024), g0, nsw_o)
    assert o["status"] == 0
    chains, burn, keep = 256, 150, 120
    eng = make_engine(chains, seed, y, X, obs, prior, ss, sig_up, g0)
    eng.ss_sweep(burn)
    acc = {k: [] for k in ("sig", "lev", "inc", "s0", "s1")}
    for _ in range(keep // 10):
        eng.ss_sweep(10)
        gam, beta, sig = eng.get_states()
        acc["sig"].append(np.log(sig))
        acc["inc"].append(gam.astype(float))
        lev, s0, s1 = [], [], []
        for c in range(0, chains, 8):
            st = eng.ss_get_state(c)
            lev.append(np.log(st["level_sigsq"]))
            s0.append(st["state"][10])
            s1.append(st["state"][T - 5])
        acc["lev"].append(np.array(lev)); acc["s0"].append(np.array(s0)); acc["s1"].append(np.array(s1))

    def batch_se(x, nb=20):
        b = np.array_split(np.asarray(x, float), nb)
        m = np.array([v.mean(0) for v in b])
        return m.std(0, ddof=1) / np.sqrt(nb)

    def device_stats(key):
        a = np.array(acc[key])            # rounds x chains[...]
        per_chain = a.mean(0)             # chains are independent: SE across chains
        return per_chain.mean(0), per_chain.std(0, ddof=1) / np.sqrt(per_chain.shape[0])

    checks = {
        "sig": np.log(o["sigsq"][burn_o:]), "lev": np.log(o["level_sigsq"][burn_o:]),
        "inc": o["gamma"][burn_o:].astype(float), "s0": o["state"][burn_o:, 10],
        "s1": o["state"][burn_o:, T - 5],
    }
    for key, series in checks.items():
        m_d, se_d = device_stats(key)
        m_o, se_o = series.mean(0), batch_se(series)
        z = np.abs(m_d - m_o) / np.sqrt(se_d ** 2 + se_o ** 2 + 1e-12)
        assert np.all(z < 5.0), (key, m_d, m_o, z)
