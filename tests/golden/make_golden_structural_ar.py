"""Generates tests/golden/ssm_*ar*.npz and kat_ar_stationary.npz: the structural
model with an ArStateModel block + ArPosteriorSampler of the COMPILED, UNMODIFIED
reference (oracle/ref_driver.cpp: ref_ssm_ar_run, ref_ar_check_stationary).  Build
container only (see make_golden.py)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import bsts_priors, structural_data, structural_spec  # noqa: E402
from make_golden import opts_kw, prior_kw, save  # noqa: E402
from oracle_lib import Ref, ssvs_options  # noqa: E402

CASES = [  # name, trend, nseasons, T, missing, seed, ar coefficients of the data
    ("ssm_level_ar1", 1, 0, 150, 0.0, 41, [0.8]),
    ("ssm_trend_seasonal4_ar2_missing", 2, 4, 160, 0.04, 42, [1.2, -0.4]),
    ("ssm_level_ar3", 1, 0, 220, 0.0, 43, [0.9, 0.3, -0.35]),
]


def main():
    R = Ref()
    p, nsw = 6, 60
    for name, trend, ns, T, miss, seed, coef in CASES:
        X, y, _, obs = structural_data(T, p, 2, ns, seed=3 + ns, missing_frac=miss, ar_coef=coef)
        prior, _, sig_up = bsts_priors(X, y, 2)
        spec = structural_spec(y, trend, ns, ar_lags=len(coef))
        opts = ssvs_options(sigma_upper_limit=sig_up)
        g0 = np.zeros(p, np.uint8)
        o = R.ssm_run(y, X, obs, prior, opts, spec, seed, g0, nsw)
        ar = spec["ar"]
        print(name, "draws with sum|phi| >= 1:", int((np.abs(o["ar_phi"]).sum(axis=1) >= 1).sum()),
              "of", nsw, "last phi", o["ar_phi"][-1], "sigsq", o["ar_sigsq"][-1])
        save(name, X=X, y=y, observed=(np.ones(T, np.uint8) if obs is None else obs),
             seed=seed, init_gamma=g0, nsweeps=nsw, trend=trend, nseasons=ns,
             var_df=spec["var_df"], var_sigma_guess=spec["var_sigma_guess"],
             var_sigma_upper_limit=spec["var_sigma_upper_limit"],
             var_initial_sigma=spec["var_initial_sigma"],
             initial_state_mean=spec["initial_state_mean"],
             initial_state_variance=spec["initial_state_variance"],
             ar_lags=ar["lags"], ar_df=ar["df"], ar_sigma_guess=ar["sigma_guess"],
             ar_sigma_upper_limit=ar["sigma_upper_limit"], ar_initial_sigma=ar["initial_sigma"],
             ar_initial_phi=ar["initial_phi"],
             gamma=o["gamma"], beta=o["beta"], sigsq=o["sigsq"], variances=o["variances"],
             ar_phi=o["ar_phi"], ar_sigsq=o["ar_sigsq"],
             state=o["state"].astype(np.float64), **prior_kw(prior), **opts_kw(opts))

    # ArModel::check_stationary on coefficient vectors around the boundary: random
    # vectors scaled so that sum |phi| is in [1, 3] (below 1 the quick bound answers)
    rng = np.random.Generator(np.random.PCG64(77))
    phis, lags, want = np.zeros((400, 8)), np.zeros(400, np.int32), np.zeros(400, np.int32)
    for i in range(400):
        L = int(rng.integers(1, 9))
        v = rng.standard_normal(L)
        v *= rng.uniform(0.9, 3.0) / np.abs(v).sum()
        phis[i, :L] = v
        lags[i] = L
        want[i] = R.lib.ref_ar_check_stationary(L, v.ctypes.data_as(C.POINTER(C.c_double)))
    assert (want >= 0).all()
    print("stationary:", int(want.sum()), "of", len(want))
    save("kat_ar_stationary", phi=phis, lags=lags, stationary=want)

    # rtrun_norm_2_mt(mu, sigma, lo, hi): the normal and the uniform envelope of the
    # interior case, and the Tn2Sampler tails (right, left through the mirror image)
    TN2 = np.array([(0.3, 0.4, -1, 1), (1.6, 0.3, -1, 1), (-2.5, 0.2, -1, 1),
                    (1.2, 0.05, -1, 1), (0.0, 5.0, -0.1, 0.1), (1.0, 0.1, -1, 1),
                    (3.0, 1.0, -1, 0.2), (-1.0, 0.3, -0.5, 1), (5, 1, -1, 1),
                    (0.99, 0.001, -1, 0.5)], dtype=np.float64)
    draws = np.zeros((len(TN2), 200))
    for i, c in enumerate(TN2):
        R._check(R.lib.ref_rng_trun_norm_2(C.c_uint64(5), *[C.c_double(v) for v in c], 200,
                                           draws[i].ctypes.data_as(C.POINTER(C.c_double))))
    save("kat_trun_norm_2", seed=5, cases=TN2, draws=draws)


if __name__ == "__main__":
    main()
