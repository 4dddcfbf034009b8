"""Generates tests/golden/poisson_*.npz: PoissonRegressionSpikeSlabSampler of the
COMPILED, UNMODIFIED reference (oracle/ref_driver.cpp: ref_poisson_run), together with
the normal-mixture approximations its NegLogGamma table yields for the counts in the
data (ref_poisson_mixtures: create_poisson_mixture_approximation_table + approximate) --
those mixtures are input DATA for the oracle and the device.  Build container only (see
make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import poisson_data, probit_slab  # noqa: E402
from make_golden import save  # noqa: E402
from oracle_lib import Ref  # noqa: E402

CASES = [  # name, n, p, signals, max exposure, intercept, seed, max_flips
    ("poisson_small_counts", 300, 10, 3, 1.0, 0.5, 31, -1),
    ("poisson_exposure", 300, 10, 3, 4.0, 1.5, 32, -1),          # counts into the tens
    ("poisson_large_counts", 200, 8, 3, 1.0, 5.0, 33, -1),       # counts of 100 - 1000: interpolated table entries
    ("poisson_p24_maxflips", 500, 24, 5, 1.0, 0.3, 34, 9),
]


def main():
    R = Ref()
    for name, n, p, nsig, mexp, icpt, seed, mf in CASES:
        X, y, ex, _ = poisson_data(n, p, nsig, seed=7 + p + int(icpt * 10), max_exposure=mexp, intercept=icpt)
        slab, pi = probit_slab(X, np.ones(n), nsig)
        g0 = np.zeros(p, np.uint8)
        g0[0] = 1
        o = R.poisson_run(X, y, ex, slab, pi, seed, g0, np.zeros(p), 40, max_flips=mf)
        mix = R.poisson_mixtures(y)
        save(name, X=X, y=y, exposure=ex, mu=slab["mu"], prec=slab["prec"], pi=pi, seed=seed,
             max_flips=mf, init_gamma=g0, nsweeps=40, gamma=o["gamma"], beta=o["beta"],
             mix_counts=mix["counts"], mix_ncomp=mix["ncomp"], mix_mu=mix["mu"], mix_sigma=mix["sigma"],
             mix_weight=mix["weight"], mix_largest_index=mix["largest_index"])
        print(name, "max count", int(y.max()), "distinct", len(mix["counts"]))


if __name__ == "__main__":
    main()
