"""Generates tests/golden/kat_forecast.npz: StateSpaceRegressionModel::
simulate_forecast of the COMPILED, UNMODIFIED reference (see make_golden.py) for
fixed parameters and final state.  Build container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from cases import state_space_data  # noqa: E402
from make_golden import save  # noqa: E402
from oracle_lib import Ref  # noqa: E402


def main():
    R = Ref()
    T, p, h, seed = 120, 6, 40, 99
    X, y, _, _ = state_space_data(T, p, 3, seed=5)
    newX = np.random.Generator(np.random.PCG64(3)).standard_normal((h, p))
    beta = np.array([3.0, 6.0, 0.0, 0.0, 0.5, 0.0])
    gamma = (beta != 0).astype(np.uint8)
    out = R.ss_forecast(y, X, beta, gamma, 0.04, 0.25, 1.7, newX, seed)
    save("kat_forecast", seed=seed, newX=newX, beta=beta, sigsq_obs=0.04, sigsq_level=0.25,
         final_state=1.7, forecast=out)


if __name__ == "__main__":
    main()
