"""Generates tests/golden/ssg_*.npz: StateSpaceRegressionModel with GENERAL lists of
state models -- any order, seasonal models with season_duration > 1 and a
time_of_first_observation, two seasonal blocks, models without a trend block, two
autoregression blocks, state dimension 53 -- sampled by the COMPILED, UNMODIFIED
reference (oracle/ref_driver.cpp: ref_ssg_run, ref_ssg_forecast).  Build container only
(see make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import bsts_priors, general_data, general_spec  # noqa: E402
from make_golden import opts_kw, prior_kw, save  # noqa: E402
from oracle_lib import Ref, ssvs_options  # noqa: E402

# name, blocks, T, data seasonals, missing, ar_coef, level in the data, seed, nsweeps, state_every
CASES = [
    ("ssg_seasonal_only", [("seasonal", 7, 1)], 120, [(7, 1)], 0.0, None, False, 41, 40, 4),
    ("ssg_ar_only", [("ar", 2)], 150, [], 0.0, [0.6, -0.3], False, 42, 40, 4),
    ("ssg_weekly_annual", [("trend",), ("seasonal", 7, 1), ("seasonal", 4, 7)], 200,
     [(7, 1), (4, 7)], 0.0, None, True, 43, 40, 4),
    ("ssg_seasonal_first_missing", [("seasonal", 4, 3), ("level",)], 150, [(4, 3)], 0.05, None,
     True, 44, 40, 4),
    ("ssg_duration_t0", [("level",), ("seasonal", 5, 4, 2)], 160, [(5, 4)], 0.0, None, True, 45,
     40, 4),
    ("ssg_two_ar", [("ar", 1), ("level",), ("ar", 2, [0.2, 0.1])], 140, [], 0.0, [0.5, 0.2], True,
     46, 40, 4),
    ("ssg_level_and_trend", [("level",), ("seasonal", 3, 2), ("trend",)], 130, [(3, 2)], 0.0,
     None, True, 47, 40, 4),
    ("ssg_big52", [("trend",), ("seasonal", 52, 7)], 420, [(52, 7)], 0.0, None, True, 48, 12, 6),
]


def spec_kw(blocks):
    """a block list as flat arrays an .npz can hold"""
    from cases import general_arrays
    kinds, ip, vpar, phi0, a0, P0 = general_arrays(blocks)
    return dict(kinds=kinds, iparams=ip, vpar=vpar, phi0=phi0, a0=a0, P0=P0)


def main():
    R = Ref()
    p = 6
    for name, desc, T, seas, miss, arc, lev, seed, nsw, every in CASES:
        X, y, _, obs = general_data(T, p, 2, seas, seed=seed + 100, missing_frac=miss,
                                    ar_coef=arc, level=lev)
        prior, _, sig_up = bsts_priors(X, y, 2)
        blocks = general_spec(y, desc)
        opts = ssvs_options(sigma_upper_limit=sig_up)
        g0 = np.zeros(p, np.uint8)
        o = R.ssg_run(y, X, obs, prior, opts, blocks, seed, g0, nsw, every)
        save(name, X=X, y=y, observed=(np.ones(T, np.uint8) if obs is None else obs),
             seed=seed, init_gamma=g0, nsweeps=nsw, state_every=every,
             gamma=o["gamma"], beta=o["beta"], sigsq=o["sigsq"], variances=o["variances"],
             phi=o["phi"], state=o["state"].astype(np.float64), **spec_kw(blocks),
             **prior_kw(prior), **opts_kw(opts))

    # simulate_forecast at fixed parameters: seasonal blocks with duration > 1 (the
    # reference simulates forecast step i with the matrices of time T - 2 + i)
    h = 30
    fc = {}
    shapes = [
        ("a", [("level",), ("seasonal", 4, 3)], 50),
        ("b", [("seasonal", 3, 5, 1), ("trend",), ("ar", 2)], 61),
        ("c", [("trend",), ("seasonal", 7, 1), ("seasonal", 4, 7)], 200),
    ]
    g = np.random.Generator(np.random.PCG64(9))
    newX = g.standard_normal((h, p))
    beta = np.array([3.0, 0.0, 0.5, 0.0, 0.0, 0.0])
    for key, desc, T in shapes:
        blocks = general_spec(np.arange(10.0), desc)
        nb = len(blocks)
        m = sum(b["dim"] for b in blocks)
        sig = 0.05 + 0.2 * g.random((nb, 2))
        phi = np.zeros((nb, 16))
        for i, b in enumerate(blocks):
            if b["kind"] == 4:
                phi[i, :b["lags"]] = [0.5, -0.2][:b["lags"]]
        fs = g.standard_normal(m)
        out = R.ssg_forecast(T, newX, beta, 0.04, blocks, sig, phi, fs, 77)
        fc[key + "_T"] = T
        fc[key + "_sigsq"] = sig
        fc[key + "_phi"] = phi
        fc[key + "_final_state"] = fs
        fc[key + "_forecast"] = out
        for k, v in spec_kw(blocks).items():
            fc[key + "_" + k] = v
    save("kat_general_forecast", seed=77, newX=newX, beta=beta, sigsq_obs=0.04,
         shapes=np.array([s[0] for s in shapes]), **fc)


if __name__ == "__main__":
    main()
