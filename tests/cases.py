"""Seeded synthetic inputs shared by the golden generator and the parity tests.

Every case is a pure function of its arguments (numpy PCG64 with an explicit
seed), so fixtures, oracle runs and GPU runs all see identical inputs.
"""
import numpy as np


def regression_data(n, p, nsignal, seed, intercept=True, noise_sd=1.0,
                    collinear=None):
    """X[:,0]=1 (if intercept), rest iid N(0,1); beta_true alternating
    +-{1,2,3} on the first nsignal columns."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((n, p))
    if intercept:
        X[:, 0] = 1.0
    if collinear:
        # columns listed in `collinear` become noisy copies of the first one
        base = collinear[0]
        for j in collinear[1:]:
            X[:, j] = X[:, base] + 0.05 * rng.standard_normal(n)
    beta = np.zeros(p)
    mags = [1.0, 2.0, 3.0]
    for j in range(nsignal):
        beta[j] = mags[j % 3] * (1 if j % 2 == 0 else -1)
    y = X @ beta + noise_sd * rng.standard_normal(n)
    return X, y, beta


def suf_from_xy(X, y):
    n = X.shape[0]
    return dict(xtx=X.T @ X, xty=X.T @ y, yty=float(y @ y), n=float(n),
                sumy=float(y.sum()), xsum=X.sum(axis=0))


def spike_slab_prior(suf, expected_model_size, kappa=0.01,
                     diagonal_shrinkage=0.5, prior_df=0.01, expected_r2=0.5,
                     force_intercept=True, prior_mean=None):
    """The R SpikeSlabPrior defaults (spike.slab.prior.R:143-155) in raw form:
    b = (ybar, 0...), Omega^{-1} = kappa*((1-w) XtX/n + w diag(XtX/n)),
    ChisqModel(df, sqrt(1-r2)*sd(y)), pi_j = ems/p (pi_0 = 1 if forced)."""
    p = len(suf["xty"])
    n = suf["n"]
    ybar = suf["sumy"] / n
    xtxn = suf["xtx"] / n
    w = diagonal_shrinkage
    ominv = kappa * ((1 - w) * xtxn + w * np.diag(np.diag(xtxn)))
    b = np.zeros(p)
    b[0] = ybar
    if prior_mean is not None:
        b = np.asarray(prior_mean, dtype=float)
    sdy = np.sqrt((suf["yty"] - n * ybar * ybar) / (n - 1))
    pi = np.full(p, min(1.0, expected_model_size / p))
    if force_intercept:
        pi[0] = 1.0
    return dict(b=b, ominv=ominv, df=prior_df,
                sigma_guess=float(np.sqrt(1 - expected_r2) * sdy), pi=pi)


def state_space_data(T, p, nsignal, seed, level_sd=0.5, obs_sd=0.2,
                     missing_frac=0.0):
    """Local level + regression: no intercept column."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((T, p))
    beta = np.zeros(p)
    for j in range(nsignal):
        beta[j] = 3.0 * (j + 1)
    level = np.cumsum(level_sd * rng.standard_normal(T))
    y = level + X @ beta + obs_sd * rng.standard_normal(T)
    observed = None
    if missing_frac > 0:
        observed = (rng.random(T) >= missing_frac).astype(np.uint8)
        observed[0] = 1
    return X, y, beta, observed


def bsts_priors(X, y, expected_model_size):
    """bsts R defaults (bsts.R:426-457, add.local.level.R:181-190)."""
    T, p = X.shape
    suf = suf_from_xy(X, y)
    sdy = float(np.std(y, ddof=1))
    prior = spike_slab_prior(suf, expected_model_size, force_intercept=False,
                             prior_mean=np.zeros(p))
    prior["sigma_guess"] = float(np.sqrt(0.5) * sdy)
    ss = dict(level_df=0.01, level_sigma_guess=0.01 * sdy,
              level_sigma_upper_limit=sdy, initial_state_mean=float(y[0]),
              initial_state_variance=sdy * sdy, initial_level_sigma=1.0)
    sigma_upper = 1.2 * sdy
    return prior, ss, sigma_upper


def structural_data(T, p, nsig, nseasons, seed, slope=0.02, missing_frac=0.0, ar_coef=None):
    """y = trend (random walk with drift) + seasonal pattern + X beta + noise
    [+ a stationary autoregression with coefficients ar_coef, innovation sd 0.5]"""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((T, p))
    beta = np.zeros(p)
    beta[:nsig] = 3.0 * (1 + np.arange(nsig))
    level = np.cumsum(slope + 0.1 * rng.standard_normal(T))
    seas = np.zeros(T)
    if nseasons > 0:
        pattern = rng.standard_normal(nseasons)
        pattern -= pattern.mean()
        seas = pattern[np.arange(T) % nseasons]
    y = level + seas + X @ beta + 0.2 * rng.standard_normal(T)
    if ar_coef is not None:
        # (drawn from a generator of its own: the cases without it keep their data)
        r2 = np.random.Generator(np.random.PCG64(seed + 1000))
        L = len(ar_coef)
        u = np.zeros(T + 50 + L)
        e = 0.5 * r2.standard_normal(T + 50 + L)
        for t in range(L, len(u)):
            u[t] = sum(ar_coef[i] * u[t - 1 - i] for i in range(L)) + e[t]
        y = y + u[-T:]
    observed = None
    if missing_frac > 0:
        observed = (rng.random(T) >= missing_frac).astype(np.uint8)
        observed[0] = 1
    return X, y, beta, observed


def structural_spec(y, trend, nseasons, ar_lags=0):
    """bsts-style defaults (add.local.linear.trend.R, add.seasonal.R, add.ar.R): sd
    priors with guess 0.01 sd(y), df 0.01, upper limit sd(y); initial state N(y[0] or
    0, sd(y)^2).  Three-element arrays: level, slope, seasonal; ar_lags > 0 adds an
    ArStateModel block (spec["ar"]) after them."""
    sdy = float(np.std(y, ddof=1))
    m = trend + (nseasons - 1 if nseasons > 0 else 0) + ar_lags
    a0 = np.zeros(m)
    a0[0] = float(y[0])
    ar = None
    if ar_lags > 0:
        ar = dict(lags=ar_lags, df=0.01, sigma_guess=0.01 * sdy, sigma_upper_limit=sdy,
                  initial_sigma=1.0, initial_phi=np.zeros(ar_lags))
    return dict(trend=trend, nseasons=nseasons, ar=ar,
                var_df=np.array([0.01, 0.01, 0.01]),
                var_sigma_guess=np.array([0.01 * sdy] * 3),
                var_sigma_upper_limit=np.array([sdy] * 3),
                var_initial_sigma=np.array([1.0, 0.5, 0.7]),
                initial_state_mean=a0,
                initial_state_variance=np.full(m, sdy * sdy))


KIND_LOCAL_LEVEL, KIND_LOCAL_LINEAR_TREND, KIND_SEASONAL, KIND_AR = 1, 2, 3, 4
KIND_STATIC_INTERCEPT, KIND_TRIG, KIND_SEMILOCAL = 5, 6, 7


def trig_rotations(period, frequencies):
    """the (cos, sin) pairs of TrigStateModel's rotation blocks as the reference computes them
    (TrigStateModel.cpp:144-147: freq = 2 * Constants::pi * f / period; C's cos / sin, which
    math.cos / math.sin call -- numpy's vector forms may differ in the last bit)"""
    import math
    out = []
    for f in frequencies:
        w = 2 * 3.141592653589793 * float(f) / float(period)
        out += [math.cos(w), math.sin(w)]
    return np.array(out)


def general_spec(y, blocks):
    """a list of state models in the order they are added (add_state), bsts-style
    defaults as structural_spec.  blocks: tuples ("level",), ("trend",), ("seasonal",
    nseasons, duration[, time_of_first_observation]), ("ar", lags[, initial_phi]),
    ("intercept",) (StaticInterceptStateModel), ("trig", period, frequencies) (TrigStateModel),
    ("semilocal"[, force_stationary[, force_positive]]) (SemilocalLinearTrendStateModel with bsts's
    AddSemilocalLinearTrend priors: slope mean N(0, sd(y)), AR(1) coefficient N(0, 1)).
    Returns the list of block dicts the oracle / reference / engine wrappers take."""
    sdy = float(np.std(y, ddof=1))
    out = []
    first = True
    init_sigma = {"level": [1.0], "trend": [1.0, 0.5], "seasonal": [0.7], "ar": [1.0], "intercept": [],
                  "trig": [0.4], "semilocal": [0.8, 0.3]}
    for b in blocks:
        name = b[0]
        kind = {"level": KIND_LOCAL_LEVEL, "trend": KIND_LOCAL_LINEAR_TREND,
                "seasonal": KIND_SEASONAL, "ar": KIND_AR, "intercept": KIND_STATIC_INTERCEPT,
                "trig": KIND_TRIG, "semilocal": KIND_SEMILOCAL}[name]
        nv = 2 if name in ("trend", "semilocal") else (0 if name == "intercept" else 1)
        d = dict(kind=kind, nseasons=0, duration=1, t0=0, lags=0,
                 df=np.full(nv, 0.01), sigma_guess=np.full(nv, 0.01 * sdy),
                 sigma_upper_limit=np.full(nv, sdy),
                 initial_sigma=np.array(init_sigma[name]), initial_phi=np.zeros(0))
        if name == "seasonal":
            d["nseasons"], d["duration"] = int(b[1]), int(b[2])
            d["t0"] = int(b[3]) if len(b) > 3 else 0
            dim = d["nseasons"] - 1
        elif name == "ar":
            d["lags"] = int(b[1])
            d["initial_phi"] = (np.asarray(b[2], float) if len(b) > 2
                                else np.zeros(d["lags"]))
            dim = d["lags"]
        elif name == "intercept":
            dim = 1
        elif name == "semilocal":
            d["force_stationary"] = int(b[1]) if len(b) > 1 else 1
            d["force_positive"] = int(b[2]) if len(b) > 2 else 0
            # slope mean prior (mu, sigma), slope AR(1) prior (mu, sigma), initial mu, initial phi
            d["slope_priors"] = np.array([0.0, sdy, 0.0, 1.0, 0.0, 0.0])
            dim = 3
        elif name == "trig":
            d["period"] = float(b[1])
            d["frequencies"] = np.asarray(b[2], float)
            d["rotations"] = trig_rotations(d["period"], d["frequencies"])
            dim = 2 * len(d["frequencies"])
        else:
            dim = nv
        a0 = np.zeros(dim)
        if first and name in ("level", "trend", "intercept", "semilocal"):
            a0[0] = float(y[0])
            first = False
        d["a0"] = a0
        d["P0"] = np.full(dim, sdy * sdy)
        if name == "semilocal":
            d["P0"][2] = 0.0      # (the slope's long-run mean is a parameter, not a draw)
        d["dim"] = dim
        out.append(d)
    return out


def general_arrays(blocks):
    """the flat arrays of ref_ssg_run / bo_ssm_add_block / ba_ss_add_state"""
    nb = len(blocks)
    kinds = np.array([b["kind"] for b in blocks], np.int32)
    ip = np.zeros((nb, 3), np.int32)
    vpar = np.ones((nb, 2, 4))
    phi0 = np.zeros((nb, 16))
    for i, b in enumerate(blocks):
        if b["kind"] == KIND_SEASONAL:
            ip[i] = (b["nseasons"], b["duration"], b["t0"])
        elif b["kind"] == KIND_AR:
            ip[i, 0] = b["lags"]
            phi0[i, :b["lags"]] = b["initial_phi"]
        elif b["kind"] == KIND_SEMILOCAL:
            ip[i, 0], ip[i, 1] = b["force_stationary"], b["force_positive"]
            phi0[i, :6] = b["slope_priors"]
        elif b["kind"] == KIND_TRIG:
            # (ref_ssg_run: the number of frequencies; the period, then the frequencies)
            nf = len(b["frequencies"])
            ip[i, 0] = nf
            if nf <= 15:     # (what fits ref_ssg_run's sixteen slots; the oracle and the engine take b["rotations"])
                phi0[i, 0] = b["period"]
                phi0[i, 1:1 + nf] = b["frequencies"]
        for v in range(len(b["df"])):
            vpar[i, v] = (b["df"][v], b["sigma_guess"][v], b["sigma_upper_limit"][v],
                          b["initial_sigma"][v])
    a0 = np.concatenate([b["a0"] for b in blocks])
    P0 = np.concatenate([b["P0"] for b in blocks])
    return kinds, ip, vpar, phi0, a0, P0


def blocks_of(g, prefix=""):
    """the block list of a golden file written by make_golden_structural_general.py"""
    kinds, ip, vpar = g[prefix + "kinds"], g[prefix + "iparams"], g[prefix + "vpar"]
    phi0, a0, P0 = g[prefix + "phi0"], g[prefix + "a0"], g[prefix + "P0"]
    out, first = [], 0
    for i, k in enumerate(kinds):
        k = int(k)
        nv = 2 if k in (KIND_LOCAL_LINEAR_TREND, KIND_SEMILOCAL) else (0 if k == KIND_STATIC_INTERCEPT else 1)
        d = dict(kind=k, nseasons=0, duration=1, t0=0, lags=0, df=vpar[i, :nv, 0],
                 sigma_guess=vpar[i, :nv, 1], sigma_upper_limit=vpar[i, :nv, 2],
                 initial_sigma=vpar[i, :nv, 3], initial_phi=np.zeros(0))
        if k == KIND_SEASONAL:
            d["nseasons"], d["duration"], d["t0"] = (int(v) for v in ip[i])
            dim = d["nseasons"] - 1
        elif k == KIND_AR:
            d["lags"] = int(ip[i, 0])
            d["initial_phi"] = phi0[i, :d["lags"]]
            dim = d["lags"]
        elif k == KIND_STATIC_INTERCEPT:
            dim = 1
        elif k == KIND_SEMILOCAL:
            d["force_stationary"], d["force_positive"] = int(ip[i, 0]), int(ip[i, 1])
            d["slope_priors"] = np.array(phi0[i, :6])
            dim = 3
        elif k == KIND_TRIG:
            nf = int(ip[i, 0])
            d["period"], d["frequencies"] = float(phi0[i, 0]), np.array(phi0[i, 1:1 + nf])
            d["rotations"] = trig_rotations(d["period"], d["frequencies"])
            dim = 2 * nf
        else:
            dim = nv
        d["a0"], d["P0"], d["dim"] = a0[first:first + dim], P0[first:first + dim], dim
        first += dim
        out.append(d)
    return out


def general_data(T, p, nsig, seasonals, seed, slope=0.02, missing_frac=0.0, ar_coef=None,
                 level=True, trig=None, intercept=0.0):
    """y = [random walk with drift] + seasonal patterns ((nseasons, duration) pairs) +
    X beta + noise [+ a stationary autoregression]"""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((T, p))
    beta = np.zeros(p)
    beta[:nsig] = 3.0 * (1 + np.arange(nsig))
    y = X @ beta + 0.2 * rng.standard_normal(T)
    if level:
        y = y + np.cumsum(slope + 0.1 * rng.standard_normal(T))
    for ns, dur in seasonals:
        pattern = rng.standard_normal(ns)
        pattern -= pattern.mean()
        y = y + pattern[(np.arange(T) // dur) % ns]
    for period, freqs in (trig or []):
        for f in freqs:
            a, b = rng.standard_normal(2)
            w = 2 * np.pi * f / period * np.arange(T)
            y = y + a * np.cos(w) + b * np.sin(w)
    y = y + intercept
    if ar_coef is not None:
        L = len(ar_coef)
        u = np.zeros(T + 50 + L)
        e = 0.5 * rng.standard_normal(T + 50 + L)
        for t in range(L, len(u)):
            u[t] = sum(ar_coef[i] * u[t - 1 - i] for i in range(L)) + e[t]
        y = y + u[-T:]
    observed = None
    if missing_frac > 0:
        observed = (rng.random(T) >= missing_frac).astype(np.uint8)
        observed[0] = 1
    return X, y, beta, observed


def probit_data(n, p, nsig, seed, max_trials=1):
    """binomial probit data: X[:, 0] = 1, successes y out of ntrials"""
    from math import erf
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((n, p))
    X[:, 0] = 1.0
    beta = np.zeros(p)
    beta[:nsig] = np.array([0.3, 1.0, -0.8, 0.6, -0.5, 0.4, 0.9, -0.7])[:nsig]
    eta = X @ beta
    prob = np.array([0.5 * (1 + erf(e / np.sqrt(2))) for e in eta])
    nt = np.ones(n) if max_trials == 1 else rng.integers(1, max_trials + 1, n).astype(float)
    y = rng.binomial(nt.astype(int), prob).astype(float)
    return X, y, nt, beta


def probit_slab(X, ntrials, expected_model_size, prior_nobs=1.0):
    """a fixed-precision slab in the style of the reference's logit / probit spike-slab
    priors: precision = prior_nobs * X'NX / n (shrunk to its diagonal by half)"""
    n, p = X.shape
    xtx = (X * ntrials[:, None]).T @ X
    prec = prior_nobs * (0.5 * np.diag(np.diag(xtx / n)) + 0.5 * xtx / n)
    pi = np.full(p, min(1.0, expected_model_size / p))
    return dict(mu=np.zeros(p), prec=prec), pi


def logit_data(n, p, nsig, seed, max_trials=1):
    """binomial logit data: X[:, 0] = 1, successes y out of ntrials"""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((n, p))
    X[:, 0] = 1.0
    beta = np.zeros(p)
    beta[:nsig] = np.array([0.4, 1.5, -1.2, 1.0, -0.8, 0.7, 1.3, -1.1])[:nsig]
    prob = 1.0 / (1.0 + np.exp(-(X @ beta)))
    nt = np.ones(n) if max_trials == 1 else rng.integers(1, max_trials + 1, n).astype(float)
    y = rng.binomial(nt.astype(int), prob).astype(float)
    return X, y, nt, beta


def poisson_data(n, p, nsig, seed, max_exposure=1.0, intercept=0.5):
    """Poisson regression data: X[:, 0] = 1, counts y with log rate x'beta + log exposure"""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((n, p))
    X[:, 0] = 1.0
    beta = np.zeros(p)
    beta[:nsig] = np.array([intercept, 0.6, -0.5, 0.4, -0.35, 0.3, 0.5, -0.45])[:nsig]
    exposure = np.ones(n) if max_exposure == 1.0 else rng.uniform(0.5, max_exposure, n)
    y = rng.poisson(exposure * np.exp(X @ beta)).astype(float)
    return X, y, exposure, beta
