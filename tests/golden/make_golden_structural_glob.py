"""Generates tests/golden/ssq_*.npz and kat_glob_forecast.npz: StateSpaceRegressionModel with the
state models round 6 adds along SURVEY 8f-2's glob -- StaticInterceptStateModel
(StaticInterceptStateModel.hpp:35) and TrigStateModel (TrigStateModel.cpp:130-223), alone and in
lists with the round-4 models -- sampled by the COMPILED, UNMODIFIED reference
(oracle/ref_driver.cpp: ref_ssg_run, ref_ssg_forecast; the samplers as bsts builds them,
Interfaces/R/bsts/src/create_state_model.cpp:559-586).  Build container only (see make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import bsts_priors, general_data, general_spec  # noqa: E402
from make_golden import opts_kw, prior_kw, save  # noqa: E402
from make_golden_structural_general import spec_kw  # noqa: E402
from oracle_lib import Ref, ssvs_options  # noqa: E402

# name, blocks, T, data seasonals, data trig, data intercept, level in the data, missing, ar, seed, nsweeps, state_every
CASES = [
    ("ssq_intercept_ar", [("intercept",), ("ar", 2)], 150, [], None, 4.0, False, 0.0, [0.6, -0.3], 61, 40, 4),
    ("ssq_intercept_seasonal_missing", [("seasonal", 4, 3), ("intercept",)], 140, [(4, 3)], None, -2.5, False,
     0.05, None, 62, 40, 4),
    ("ssq_trig_only", [("trig", 12.0, [1.0, 2.0])], 130, [], [(12.0, [1.0, 2.0])], 0.0, False, 0.0, None, 63,
     40, 4),
    ("ssq_trend_trig", [("trend",), ("trig", 7.0, [1.0, 2.0, 3.0])], 160, [], [(7.0, [1.0, 2.0, 3.0])], 0.0, True,
     0.03, None, 64, 40, 4),
    ("ssq_trig_level_seasonal", [("trig", 30.5, [1.0]), ("level",), ("seasonal", 7, 1)], 150, [(7, 1)],
     [(30.5, [1.0])], 0.0, True, 0.0, None, 65, 40, 4),
    ("ssq_two_trig_intercept", [("intercept",), ("trig", 24.0, [1.0, 3.0]), ("trig", 5.0, [2.0])], 120, [],
     [(24.0, [1.0, 3.0]), (5.0, [2.0])], 1.5, False, 0.0, None, 66, 40, 4),
    # SemilocalLinearTrendStateModel: stationary slope (phi in [-1, 1]), alone; phi in [0, 1], with a
    # seasonal block ahead of it and missing observations; no truncation at all, beside a trig block
    ("ssq_semilocal", [("semilocal",)], 150, [], None, 0.0, True, 0.0, None, 67, 40, 4),
    ("ssq_seasonal_semilocal_missing", [("seasonal", 7, 1), ("semilocal", 1, 1)], 160, [(7, 1)], None, 0.0, True,
     0.04, None, 68, 40, 4),
    ("ssq_semilocal_free_trig", [("semilocal", 0, 0), ("trig", 12.0, [1.0])], 140, [], [(12.0, [1.0])], 0.0, True,
     0.0, None, 69, 40, 4),
]


def main():
    R = Ref()
    p = 6
    for name, desc, T, seas, trig, icpt, lev, miss, arc, seed, nsw, every in CASES:
        X, y, _, obs = general_data(T, p, 2, seas, seed=seed + 100, missing_frac=miss, ar_coef=arc, level=lev,
                                    trig=trig, intercept=icpt)
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

    # simulate_forecast at fixed parameters
    h = 24
    fc = {}
    shapes = [
        ("a", [("intercept",), ("trig", 12.0, [1.0, 2.0])], 50),
        ("b", [("trig", 7.0, [1.0, 2.0, 3.0]), ("trend",), ("seasonal", 3, 5, 1)], 61),
        ("c", [("seasonal", 4, 2), ("semilocal",)], 40),
    ]
    g = np.random.Generator(np.random.PCG64(19))
    newX = g.standard_normal((h, p))
    beta = np.array([3.0, 0.0, 0.5, 0.0, 0.0, 0.0])
    from cases import general_arrays
    for key, desc, T in shapes:
        blocks = general_spec(np.arange(10.0), desc)
        nb = len(blocks)
        m = sum(b["dim"] for b in blocks)
        sig = 0.05 + 0.2 * g.random((nb, 2))
        for i, b in enumerate(blocks):
            if b["kind"] == 5:
                sig[i] = 0.0
        for b in blocks:
            if b["kind"] == 7:
                b["slope_priors"][4:] = (0.07, 0.8)      # (the slope's mu and phi of this forecast)
        phi = general_arrays(blocks)[3]      # (a trig block's period and frequencies, a semilocal trend's mu and phi ride there)
        fs = g.standard_normal(m)
        out = R.ssg_forecast(T, newX, beta, 0.04, blocks, sig, phi, fs, 78)
        fc[key + "_T"] = T
        fc[key + "_sigsq"] = sig
        fc[key + "_phi"] = phi
        fc[key + "_final_state"] = fs
        fc[key + "_forecast"] = out
        for k, v in spec_kw(blocks).items():
            fc[key + "_" + k] = v
    save("kat_glob_forecast", seed=78, newX=newX, beta=beta, sigsq_obs=0.04,
         shapes=np.array([s[0] for s in shapes]), **fc)


if __name__ == "__main__":
    main()
