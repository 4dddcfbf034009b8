"""Generates tests/golden/adaptive_*.npz -- AdaptiveSpikeSlabRegressionSampler
draws of the COMPILED, UNMODIFIED reference (oracle/_ref/libboomref.so; see
make_golden.py).  Build container only:  python tests/golden/make_golden_adaptive.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import regression_data, spike_slab_prior, suf_from_xy  # noqa: E402
from make_golden import prior_kw, save  # noqa: E402
from oracle_lib import Ref, ssvs_options  # noqa: E402

CASES = {
    # name: (n, p, nsignal, data seed, data kw, sampler options)
    "adaptive_c1": (1000, 20, 6, 1, {}, {}),
    "adaptive_p150": (600, 150, 12, 2, {}, {}),
    "adaptive_collinear": (400, 40, 4, 6, dict(collinear=[1, 7, 9, 20]), {}),
    "adaptive_options": (600, 150, 12, 5, {},
                         dict(max_flips=30, step_size=0.05, target=0.2, max_model_size=15,
                              sigma_upper_limit=20.0)),
}


def main():
    R = Ref()
    seed, nsweeps = 8675309, 120
    for name, (n, p, nsig, dseed, dkw, so) in CASES.items():
        X, y, _ = regression_data(n, p, nsig, seed=dseed, **dkw)
        suf = suf_from_xy(X, y)
        prior = spike_slab_prior(suf, nsig)
        g0 = np.zeros(p, np.uint8)
        g0[0] = 1
        opts = ssvs_options(max_model_size=so.get("max_model_size", -1),
                            sigma_upper_limit=so.get("sigma_upper_limit", float("inf")))
        mf, st, tg = so.get("max_flips", -1), so.get("step_size", -1.0), so.get("target", -1.0)
        r = R.adaptive_run(suf, prior, opts, seed, g0, nsweeps, mf, st, tg)
        save(name, seed=seed, xtx=suf["xtx"], xty=suf["xty"], yty=suf["yty"], n=suf["n"],
             sumy=suf["sumy"], xsum=suf["xsum"], init_gamma=g0,
             opt_max_model_size=opts["max_model_size"],
             opt_sigma_upper_limit=opts["sigma_upper_limit"], max_flips=mf, step_size=st,
             target=tg, gamma=np.packbits(r["gamma"], axis=1), beta=r["beta"],
             sigsq=r["sigsq"], **prior_kw(prior))


if __name__ == "__main__":
    main()
