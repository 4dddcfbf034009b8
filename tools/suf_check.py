#!/usr/bin/env python3
"""Diagnostic: the device X'X build (MFMA, row slices) against numpy for a
sequence of changing shapes on one engine."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, boom_amd
from cases import regression_data
eng = boom_amd.Engine(1)
for (n, p, seed) in [(1000, 20, 1), (333, 71, 21), (333, 71, 21), (1000, 20, 1), (500, 130, 3), (333, 71, 21), (64, 5, 2), (10000, 512, 4)]:
    X, y, _ = regression_data(n, p, 5 if p > 5 else 2, seed=seed)
    eng.build_suf_from_xy(X, y)
    s = eng.get_suf()
    ref = X.T @ X
    err = np.max(np.abs(s["xtx"] - ref) / np.maximum(np.abs(ref), 1e-6))
    print(n, p, "max rel err %.3g" % err, "sym", np.array_equal(s["xtx"], s["xtx"].T))
