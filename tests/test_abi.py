"""CPU-side checks of the drop-in boundary: the shared library loads without a
GPU and exports exactly the symbols include/boom_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "boom_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ba_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import boom_amd
    lib = boom_amd.load_library()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    # and the python plumbing binds each of them with a signature
    from boom_amd.capi import SIGNATURES
    assert sorted(SIGNATURES) == syms


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import boom_amd
    with pytest.raises(boom_amd.BoomAmdError):
        boom_amd.Engine(4)


def test_product_does_not_touch_the_oracle():
    """Nothing under boom_amd/ may import, link or call oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "boom_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "boom_oracle" not in txt and "oracle_lib" not in txt, f
                assert "libboomref" not in txt, f


def test_pybind_module_imports_and_mirrors_the_boom_names():
    """boom_amd._boom: the BayesBoom-shaped names of the path (no GPU needed to
    import it or to build the prior objects)"""
    import numpy as np
    import boom_amd._boom as boom
    for name in ("RegressionModel", "BregVsSampler", "MvnGivenScalarSigma", "ChisqModel",
                 "VariableSelectionPrior", "PosteriorSampler", "GlmCoefs"):
        assert hasattr(boom, name), name
    slab = boom.MvnGivenScalarSigma(np.zeros(3), np.eye(3))
    assert slab.dim == 3
    assert boom.ChisqModel(2.0, 1.5).sigma == 1.5
    assert boom.VariableSelectionPrior(np.full(3, 0.5)).potential_nvars == 3
