"""Generates tests/golden/ssm_*.npz: StateSpaceRegressionModel + trend (local
level / local linear trend with independent variance samplers) + seasonal state,
sampled by the COMPILED, UNMODIFIED reference (oracle/ref_driver.cpp:
ref_ssm_run).  Build container only (see make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import bsts_priors, structural_data, structural_spec  # noqa: E402
from make_golden import opts_kw, prior_kw, save  # noqa: E402
from oracle_lib import Ref, ssvs_options  # noqa: E402

CASES = [  # name, trend, nseasons, T, missing, seed
    ("ssm_level", 1, 0, 120, 0.0, 31),
    ("ssm_trend", 2, 0, 120, 0.0, 32),
    ("ssm_level_seasonal7", 1, 7, 150, 0.0, 33),
    ("ssm_trend_seasonal4_missing", 2, 4, 150, 0.05, 34),
    ("ssm_trend_seasonal12", 2, 12, 200, 0.0, 35),
]


def main():
    R = Ref()
    p, nsw = 6, 60
    for name, trend, ns, T, miss, seed in CASES:
        X, y, _, obs = structural_data(T, p, 2, ns, seed=3 + ns, missing_frac=miss)
        prior, _, sig_up = bsts_priors(X, y, 2)
        spec = structural_spec(y, trend, ns)
        opts = ssvs_options(sigma_upper_limit=sig_up)
        g0 = np.zeros(p, np.uint8)
        o = R.ssm_run(y, X, obs, prior, opts, spec, seed, g0, nsw)
        save(name, X=X, y=y, observed=(np.ones(T, np.uint8) if obs is None else obs),
             seed=seed, init_gamma=g0, nsweeps=nsw, trend=trend, nseasons=ns,
             var_df=spec["var_df"], var_sigma_guess=spec["var_sigma_guess"],
             var_sigma_upper_limit=spec["var_sigma_upper_limit"],
             var_initial_sigma=spec["var_initial_sigma"],
             initial_state_mean=spec["initial_state_mean"],
             initial_state_variance=spec["initial_state_variance"],
             gamma=o["gamma"], beta=o["beta"], sigsq=o["sigsq"], variances=o["variances"],
             state=o["state"].astype(np.float64), **prior_kw(prior), **opts_kw(opts))

    # simulate_forecast with fixed parameters and final state
    h = 40
    fc = {}
    for trend, ns in ((2, 0), (1, 7), (2, 12)):
        X, y, _, _ = structural_data(80, p, 2, ns, seed=3)
        newX = np.random.Generator(np.random.PCG64(3)).standard_normal((h, p))
        m = trend + max(ns - 1, 0)
        beta = np.array([3.0, 0.0, 0.5, 0.0, 0.0, 0.0])
        fs = np.random.Generator(np.random.PCG64(4)).standard_normal(m)
        sig = np.array([0.2, 0.05, 0.1])
        out = R.ssm_forecast(y, X, beta, (beta != 0).astype(np.uint8), 0.04, trend, ns, sig, fs,
                             newX, 99)
        key = "t%d_s%d" % (trend, ns)
        fc[key + "_final_state"] = fs
        fc[key + "_forecast"] = out
    save("kat_structural_forecast", seed=99, newX=newX, beta=beta, sigsq_obs=0.04, sigsq=sig,
         shapes=np.array([(2, 0), (1, 7), (2, 12)]), **fc)


if __name__ == "__main__":
    main()
