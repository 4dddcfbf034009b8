"""Runs the C++ test of the BOOM-shaped host side (tests/cpp/facade_test.cpp,
built by __graft_entry__.build() / `make -C tests/cpp`) on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cpp_facade_acceptance_tests():
    exe = os.path.join(HERE, "cpp", "build", "facade_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout
