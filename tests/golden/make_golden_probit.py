"""Generates tests/golden/probit_*.npz: BinomialProbitSpikeSlabSampler of the
COMPILED, UNMODIFIED reference (oracle/ref_driver.cpp: ref_probit_run), and
rtrun_norm_mt known answers.  Build container only (see make_golden.py).

The sampler carries continuous latent data from sweep to sweep, so rounding
differences between two implementations grow (about 3x per sweep here): the
fixtures hold 12 sweeps."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import probit_data, probit_slab  # noqa: E402
from make_golden import save  # noqa: E402
from oracle_lib import Ref  # noqa: E402

CASES = [  # name, n, p, signals, max trials, clt threshold, seed
    ("probit_bernoulli", 300, 10, 3, 1, 5, 21),
    ("probit_binomial8_clt3", 300, 10, 3, 8, 3, 22),
    ("probit_binomial12", 250, 12, 4, 12, 5, 23),
]
TN = np.array([(0.3, 1.0, 0.0, 1), (0.3, 1.0, 0.0, 0), (-2.5, 1.0, 0.0, 1), (4.0, 1.0, 0.0, 0),
               (0.0, 2.0, 3.0, 1), (0.01, 1.0, 0.0, 0)])


def main():
    R = Ref()
    save("kat_trun_norm", seed=77, cases=TN,
         draws=np.stack([R.trun_norms(77, mu, sg, cut, int(ab), 512) for mu, sg, cut, ab in TN]))
    for name, n, p, nsig, mt, clt, seed in CASES:
        X, y, nt, _ = probit_data(n, p, nsig, seed=5 + mt, max_trials=mt)
        slab, pi = probit_slab(X, nt, nsig)
        g0 = np.zeros(p, np.uint8)
        g0[0] = 1
        o = R.probit_run(X, y, nt, slab, pi, seed, g0, np.zeros(p), 12, clt_threshold=clt)
        save(name, X=X, y=y, ntrials=nt, mu=slab["mu"], prec=slab["prec"], pi=pi, seed=seed,
             clt_threshold=clt, init_gamma=g0, nsweeps=12, gamma=o["gamma"], beta=o["beta"])


if __name__ == "__main__":
    main()
