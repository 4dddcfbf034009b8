"""Generates tests/golden/logit_*.npz: BinomialLogitSpikeSlabSampler of the
COMPILED, UNMODIFIED reference (oracle/ref_driver.cpp: ref_logit_run).  Build
container only (see make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import logit_data, probit_slab  # noqa: E402
from make_golden import save  # noqa: E402
from oracle_lib import Ref  # noqa: E402

CASES = [  # name, n, p, signals, max trials, clt threshold, seed, max_flips
    ("logit_bernoulli", 300, 10, 3, 1, 5, 21, -1),
    ("logit_binomial4", 300, 10, 3, 4, 5, 22, -1),
    ("logit_bernoulli_p24_maxflips", 500, 24, 5, 1, 5, 23, 9),
    # trial counts up to 60 with clt_threshold 5: most observations take
    # BinomialLogitCltDataImputer::impute_large_sample (multinomial counts by BTPE /
    # inversion binomials, then one normal draw)
    ("logit_binomial60_large_sample", 250, 8, 3, 60, 5, 24, -1),
    ("logit_binomial200_large_sample", 120, 6, 2, 200, 10, 25, -1),
]


def main():
    R = Ref()
    for name, n, p, nsig, mt, clt, seed, mf in CASES:
        X, y, nt, _ = logit_data(n, p, nsig, seed=5 + mt + p, max_trials=mt)
        slab, pi = probit_slab(X, nt, nsig)
        g0 = np.zeros(p, np.uint8)
        g0[0] = 1
        o = R.logit_run(X, y, nt, slab, pi, seed, g0, np.zeros(p), 60, clt_threshold=clt,
                        max_flips=mf)
        save(name, X=X, y=y, ntrials=nt, mu=slab["mu"], prec=slab["prec"], pi=pi, seed=seed,
             clt_threshold=clt, max_flips=mf, init_gamma=g0, nsweeps=60, gamma=o["gamma"],
             beta=o["beta"])


if __name__ == "__main__":
    main()
