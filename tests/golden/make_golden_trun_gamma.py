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

from make_golden import save  # noqa: E402
from oracle_lib import Ref  # noqa: E402

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
    save("kat_trun_gamma", seed=seed, cases=CASES, draws=out)


if __name__ == "__main__":
    main()
