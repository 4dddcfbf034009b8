"""Generates tests/golden/kat_trun_gamma.npz: rtrun_gamma_mt of the COMPILED,
UNMODIFIED reference (distributions/trun_gamma.cpp:74-100) in all three of its
regimes -- rejection from the untruncated gamma (cut < mode), the bounded
adaptive rejection sampler (cut >= mode, a > 1) and the slice sampler
(a <= 1).  Build container only (see make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import (bsts_priors, regression_data, spike_slab_prior,  # noqa: E402
                   state_space_data, suf_from_xy)
from make_golden import opts_kw, prior_kw, save  # noqa: E402
from oracle_lib import Ref, ssvs_options  # noqa: E402

# (a, b, cut): shape, rate, lower truncation point
CASES = np.array([
    (10.0, 2.0, 3.0),      # rejection, cut below the mode 4.5
    (10.0, 2.0, 4.6),      # ARS, cut just right of the mode
    (10.0, 2.0, 12.0),     # ARS, deep in the tail
    (500.5, 40.0, 14.0),   # ARS, sigma^2-draw sized shape (n/2) with a tight upper limit on sigma
    (1.5, 0.01, 1000.0),   # ARS, small shape
    (1.0, 3.0, 2.0),       # slice (a == 1: an exponential tail)
    (0.505, 0.02, 25.0),   # slice, a < 1 (small-sample level-variance draw)
    (0.505, 0.02, 1e-3),   # slice, cut near zero
])


def main():
    R = Ref()
    seed, n = 20260, 256
    out = np.stack([R.trun_gammas(seed, a, b, cut, n) for a, b, cut in CASES])
    # plain gamma draws of shape < 0.3 (rloggamma_small_alpha)
    small = np.array([0.005, 0.1, 0.25, 0.29])
    save("kat_trun_gamma", seed=seed, cases=CASES, draws=out, small_shapes=small,
         small_rate=3.0, small_draws=np.stack([R.gammas(seed, a, 3.0, 512) for a in small]))

    # BregVsSampler with a sigma upper limit below / around the residual sd: every
    # / some sigma^2 draws come from the adaptive rejection sampler
    X, y, _ = regression_data(400, 24, 4, seed=12)
    prior = spike_slab_prior(suf_from_xy(X, y), 4)
    g0 = np.zeros(24, np.uint8)
    g0[0] = 1
    for name, limit in (("ssvs_tight_sigma", 0.9), ("ssvs_binding_sigma", 1.0)):
        opts = ssvs_options(sigma_upper_limit=limit)
        suf = R.neregsuf(X, y)
        seeds = [9, 10]
        outs = [R.ssvs_run(X, y, None, prior, opts, s, g0, 150) for s in seeds]
        save(name, X=X, y=y, xtx=suf["xtx"], xty=suf["xty"], yty=suf["yty"],
             ybar=suf["ybar"], xbar=suf["xbar"], seeds=np.array(seeds),
             init_gamma=g0, nsweeps=150,
             gamma=np.stack([o["gamma"] for o in outs]),
             beta=np.stack([o["beta"] for o in outs]),
             sigsq=np.stack([o["sigsq"] for o in outs]),
             **prior_kw(prior), **opts_kw(opts))

    # bsts local level + regression on a 3-point series: shape (T - 1 + 0.01) / 2
    # <= 1 sends every level-variance draw to the slice sampler
    T, p = 3, 4
    X, y, _, obs = state_space_data(T, p, 2, seed=203)
    prior, ss, sig_up = bsts_priors(X, y, 2)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    o = R.ss_run(y, X, obs, prior, opts, ss, 23, g0, 100)
    save("ss_t3", X=X, y=y, observed=np.ones(T, np.uint8), seed=23, init_gamma=g0,
         nsweeps=100, ss_keys=np.array(sorted(ss.keys())),
         ss_vals=np.array([ss[k] for k in sorted(ss.keys())]),
         gamma=o["gamma"], beta=o["beta"], sigsq=o["sigsq"],
         level_sigsq=o["level_sigsq"], state=o["state"],
         **prior_kw(prior), **opts_kw(opts))

    # a single observation, no upper limits, level prior df 0.5: the level
    # variance is an untruncated gamma draw of shape 0.25 (rloggamma_small_alpha)
    T, p = 1, 3
    X, y, _, _ = state_space_data(8, p, 2, seed=201)
    prior, ss, _ = bsts_priors(X, y, 2)
    X, y = X[:T], y[:T]
    ss = dict(ss, level_df=0.5, level_sigma_upper_limit=np.inf)
    opts = ssvs_options(sigma_upper_limit=np.inf)
    g0 = np.zeros(p, np.uint8)
    o = R.ss_run(y, X, None, prior, opts, ss, 29, g0, 100)
    save("ss_t1", X=X, y=y, observed=np.ones(T, np.uint8), seed=29, init_gamma=g0,
         nsweeps=100, ss_keys=np.array(sorted(ss.keys())),
         ss_vals=np.array([ss[k] for k in sorted(ss.keys())]),
         gamma=o["gamma"], beta=o["beta"], sigsq=o["sigsq"],
         level_sigsq=o["level_sigsq"], state=o["state"],
         **prior_kw(prior), **opts_kw(opts))


if __name__ == "__main__":
    main()

