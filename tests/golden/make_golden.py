"""Generates tests/golden/*.npz from the COMPILED, UNMODIFIED reference
(oracle/_ref/libboomref.so, built by `make -C oracle ref` from /root/reference).

Run in the build container only:  python tests/golden/make_golden.py

Every fixture holds the inputs and the reference's outputs for one case; the
tests re-run the oracle (MT19937-64 engine, same seeds) against them, here and
on the GPU box.  Fixtures are data only -- no reference source text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import (bsts_priors, regression_data, spike_slab_prior,  # noqa: E402
                   state_space_data, suf_from_xy)
from oracle_lib import Ref, ssvs_options  # noqa: E402


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def prior_kw(prior):
    return dict(prior_b=prior["b"], prior_ominv=prior["ominv"],
                prior_df=prior["df"], prior_sigma_guess=prior["sigma_guess"],
                prior_pi=prior["pi"])


def opts_kw(o):
    return dict(opt_max_model_size=o["max_model_size"],
                opt_sigma_upper_limit=o["sigma_upper_limit"],
                opt_swap_threshold=o["swap_threshold"],
                opt_max_flips=o["max_flips"])


def main():
    R = Ref()
    seed = 8675309

    # ---- RNG known-answer sequences (mt19937_64 seeded directly) ----------
    save("kat_rng",
         seed=seed,
         uniform=R.uniforms(seed, 512),
         seed_rng=R.seed_rngs(seed, 32),
         norm=R.norms(seed, 2048),
         exp=R.exps(seed, 1024),
         gamma_shapes=np.array([0.5, 1.0, 2.5, 7.0, 505.0, 5000.005]),
         gamma_rate=3.0,
         gamma=np.stack([R.gammas(seed, a, 3.0, 1024)
                         for a in (0.5, 1.0, 2.5, 7.0, 505.0, 5000.005)]),
         trun_gamma_args=np.array([50.0, 40.0, 1.0]),
         trun_gamma=R.trun_gammas(seed, 50.0, 40.0, 1.0, 1024),
         random_int=R.random_ints(seed, 0, 511, 1024),
         shuffle=R.shuffles(seed, 512, 4),
         rmulti_prob=np.array([.1, .5, .2, .7]),
         rmulti=R.rmultis(seed, np.array([.1, .5, .2, .7]), 512))

    # ---- LinAlg known answers (SpdMatrix chol/logdet/solve/Mdist) -----------
    rng = np.random.Generator(np.random.PCG64(99))
    mats, chols, logdets, rhss, sols, xs, mds = [], [], [], [], [], [], []
    for n in (1, 2, 3, 5, 17, 40):
        B = rng.standard_normal((n + 3, n))
        A = B.T @ B + 0.1 * np.eye(n)
        rhs = rng.standard_normal(n)
        x = rng.standard_normal(n)
        L, ok = R.chol(A)
        assert ok
        mats.append(A.ravel())
        chols.append(L.ravel())
        logdets.append(R.logdet(A)[0])
        rhss.append(rhs)
        sols.append(R.solve(A, rhs)[0])
        xs.append(x)
        mds.append(R.mdist(A, x))
    notpd = np.array([[1.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    save("kat_linalg", sizes=np.array([1, 2, 3, 5, 17, 40]),
         A=np.concatenate(mats), L=np.concatenate(chols),
         logdet=np.array(logdets), rhs=np.concatenate(rhss),
         sol=np.concatenate(sols), x=np.concatenate(xs), mdist=np.array(mds),
         notpd=notpd, notpd_logdet_ok=int(R.logdet(notpd)[1]),
         notpd_solve_ok=int(R.solve(notpd, np.ones(3))[1]))

    # ---- SSVS sweeps ----------------------------------------------------------
    def ssvs_case(name, X, y, prior, opts, seeds, g0, nsweeps):
        suf = R.neregsuf(X, y)
        outs = [R.ssvs_run(X, y, None, prior, opts, s, g0, nsweeps)
                for s in seeds]
        save(name, X=X, y=y, xtx=suf["xtx"], xty=suf["xty"], yty=suf["yty"],
             ybar=suf["ybar"], xbar=suf["xbar"], seeds=np.array(seeds),
             init_gamma=g0, nsweeps=nsweeps,
             gamma=np.stack([o["gamma"] for o in outs]),
             beta=np.stack([o["beta"] for o in outs]),
             sigsq=np.stack([o["sigsq"] for o in outs]),
             **prior_kw(prior), **opts_kw(opts))

    # C1: n=1000, p=20 (regression_spike_slab_test.cc:23-59 sizes)
    X, y, _ = regression_data(1000, 20, 6, seed=1)
    prior = spike_slab_prior(suf_from_xy(X, y), 5)
    g0 = np.zeros(20, np.uint8)
    g0[0] = 1
    ssvs_case("ssvs_c1", X, y, prior, ssvs_options(), [1, seed], g0, 200)

    # log_model_prob of arbitrary models on the C1 data
    suf = R.neregsuf(X, y)
    pr2 = spike_slab_prior(suf, 5, force_intercept=False)
    G = (np.random.default_rng(0).random((64, 20)) < 0.3).astype(np.uint8)
    G[0] = 0
    save("kat_log_model_prob", xtx=suf["xtx"], xty=suf["xty"], yty=suf["yty"],
         n=suf["n"], ybar=suf["ybar"], xbar=suf["xbar"], gammas=G,
         logp=R.log_model_prob(suf, pr2, G),
         logp_max3=R.log_model_prob(suf, pr2, G, max_model_size=3),
         **prior_kw(pr2))

    # logpri() at arbitrary states on the C1 data (a prior with non-zero means)
    rs = np.random.default_rng(4)
    pr3 = spike_slab_prior(suf, 5, force_intercept=False, prior_mean=0.3 * rs.standard_normal(20))
    Gp = (rs.random((32, 20)) < 0.3).astype(np.uint8)
    Gp[0] = 0
    Bp = rs.standard_normal((32, 20)) * Gp
    Sp = np.exp(rs.normal(0, 0.5, 32))
    save("kat_logpri", xtx=suf["xtx"], xty=suf["xty"], yty=suf["yty"],
         n=suf["n"], ybar=suf["ybar"], xbar=suf["xbar"], gammas=Gp, betas=Bp, sigsqs=Sp,
         logpri=R.logpri(suf, pr3, Gp, Bp, Sp), **prior_kw(pr3))

    # convenience ctors #1 / #2 on the C1 data
    o1 = R.ssvs_run_ctor(1, X, y, [1.0, 0.5, 3.0, 0, 0], 1, ssvs_options(), 9,
                         g0, 100)
    o2 = R.ssvs_run_ctor(2, X, y, [1.0, 1.5, 0.5, 0.3, 0.25], 1,
                         ssvs_options(), 9, g0, 100)
    save("ssvs_ctors", X=X, y=y, seed=9, init_gamma=g0, nsweeps=100,
         ctor1_args=np.array([1.0, 0.5, 3.0]), ctor1_flag=1,
         ctor2_args=np.array([1.0, 1.5, 0.5, 0.3, 0.25]), ctor2_flag=1,
         gamma1=o1["gamma"], beta1=o1["beta"], sigsq1=o1["sigsq"],
         gamma2=o2["gamma"], beta2=o2["beta"], sigsq2=o2["sigsq"])

    # p = 64, kbar ~ 12
    X, y, _ = regression_data(600, 64, 12, seed=2)
    prior = spike_slab_prior(suf_from_xy(X, y), 12)
    g0 = np.zeros(64, np.uint8)
    g0[0] = 1
    ssvs_case("ssvs_p64", X, y, prior, ssvs_options(), [7], g0, 200)

    # perfectly-collinear-ish columns: exercises the correlation swap move
    # (cf. regression_spike_slab_test.cc:207-257)
    X, y, _ = regression_data(500, 30, 2, seed=3, collinear=[1, 2, 3, 7])
    prior = spike_slab_prior(suf_from_xy(X, y), 3)
    g0 = np.zeros(30, np.uint8)
    g0[0] = 1
    ssvs_case("ssvs_collinear", X, y, prior, ssvs_options(), [11], g0, 400)

    # general: non-zero prior mean everywhere, max_model_size, truncated sigma,
    # low swap threshold, no forced intercept, several variables start included
    X, y, _ = regression_data(40, 8, 3, seed=4)
    prior = spike_slab_prior(suf_from_xy(X, y), 3, prior_mean=np.linspace(-1, 1, 8),
                             force_intercept=False)
    g0 = np.zeros(8, np.uint8)
    g0[:3] = 1
    ssvs_case("ssvs_general", X, y, prior,
              ssvs_options(max_model_size=4, sigma_upper_limit=1.08,
                           swap_threshold=0.1), [5], g0, 500)
    ssvs_case("ssvs_maxflips", X, y, prior, ssvs_options(max_flips=5), [5], g0,
              300)

    # pure-noise response, no forced intercept: the chain visits the EMPTY model
    # (k = 0 closed form; set_inc zeroes excluded coefficients)
    X, y, _ = regression_data(200, 6, 0, seed=8, intercept=False)
    prior = spike_slab_prior(suf_from_xy(X, y), 1, force_intercept=False,
                             prior_mean=np.zeros(6))
    ssvs_case("ssvs_empty", X, y, prior, ssvs_options(), [3], np.zeros(6, np.uint8), 200)

    # ---- SpikeSlabSampler: the sigma^2-conditional sweep (a11) ----------------
    rng = np.random.Generator(np.random.PCG64(5))
    for kind in (0, 1):
        for case in range(2):
            n, p = (400, 24) if case == 0 else (150, 10)
            X, y, _ = regression_data(n, p, 5 if case == 0 else 3, seed=30 + case)
            w = rng.uniform(0.3, 2.0, n) if case == 0 else np.ones(n)
            xtx, xty = R.weighted_suf(X, y, w)
            mu = np.zeros(p)
            mu[0] = y.mean()
            if case == 1:
                mu = np.linspace(-0.5, 0.5, p)   # non-zero prior mean everywhere
            prec = 0.01 * (0.5 * np.diag(np.diag(xtx / n)) + 0.5 * xtx / n)
            pi = np.full(p, 0.25)
            if case == 0:
                pi[0] = 1.0
            sig = np.exp(rng.normal(0, 0.2, 150))
            g0 = np.zeros(p, np.uint8)
            g0[0] = 1
            mms, mf = (4, 6) if case == 1 else (-1, -1)
            o = R.sss_run(X, y, w, kind, mu, prec, pi, 77, g0, sig, max_model_size=mms,
                          max_flips=mf)
            save("sss_kind%d_case%d" % (kind, case), X=X, y=y, w=w, xtx=xtx, xty=xty,
                 slab_kind=kind, mu=mu, prec=prec, pi=pi, seed=77, init_gamma=g0,
                 sigsq=sig, max_model_size=mms, max_flips=mf, gamma=o["gamma"],
                 beta=o["beta"])

    # ---- state space (local level + regression) -----------------------------
    for name, miss, sd in (("ss_t200", 0.0, 22), ("ss_t200_missing", 0.05, 21)):
        X, y, _, obs = state_space_data(200, 8, 3, seed=5, missing_frac=miss)
        prior, ss, sig_up = bsts_priors(X, y, 3)
        opts = ssvs_options(sigma_upper_limit=sig_up)
        g0 = np.zeros(8, np.uint8)
        o = R.ss_run(y, X, obs, prior, opts, ss, sd, g0, 100)
        save(name, X=X, y=y,
             observed=(np.ones(200, np.uint8) if obs is None else obs),
             seed=sd, init_gamma=g0, nsweeps=100,
             ss_keys=np.array(sorted(ss.keys())),
             ss_vals=np.array([ss[k] for k in sorted(ss.keys())]),
             gamma=o["gamma"], beta=o["beta"], sigsq=o["sigsq"],
             level_sigsq=o["level_sigsq"], state=o["state"],
             **prior_kw(prior), **opts_kw(opts))

    # one isolated impute_state (filter + simulation smoother + suf update)
    X, y, _, obs = state_space_data(200, 8, 3, seed=5, missing_frac=0.05)
    beta = np.array([3, 6, 9, 0, 0, 0, 0, 0.])
    gam = (beta != 0).astype(np.uint8)
    o = R.ss_impute_state(y, X, obs, beta, gam, 0.04, 0.25, float(y[0]), 4.0, 77)
    save("kat_impute_state", X=X, y=y, observed=obs, beta=beta, gamma=gam,
         sigsq_obs=0.04, sigsq_level=0.25, a0=float(y[0]), P0=4.0, seed=77,
         state=o["state"], xty=o["xty"], yty=o["yty"], n=o["n"],
         level_sumsq=o["level_sumsq"], level_n=o["level_n"])


if __name__ == "__main__":
    main()
