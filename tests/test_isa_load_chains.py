"""A guard on what the compiler does with the kernels' loads: two
translation units that once had memory round trips in a row in their hot loops -- the logit
request product (four per 16-row step) and the natural-layout Kalman kernel (eight per chunk)
-- and the bsts round kernel (eight per block of a table fill, ten in the state draw) are
compiled to assembly and read by tools/isa_serial_loads.py: no chain of three (round kernel:
five) or more global loads that are each waited for before the next is issued.  CPU only (hipcc
cross-compiles); the larger kernels take minutes to compile and are audited by hand with the
same tool."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
# (unit, extra flags as the Makefile has them, shortest chain that fails the test: the round kernel
# keeps two short ones that are what the source says -- the prologue's status -> owed sweeps ->
# tag reads, three in a row, and four reads of a diagnostic record that no product run writes)
@pytest.mark.parametrize("unit,flags,min_chain", [("xtwx_cols_kernel", [], 3), ("kalman_kernel", [], 3),
                                                  ("ss_round_kernel", ["-mllvm", "-disable-machine-licm"], 5)])
def test_no_chain_of_dependent_global_loads(tmp_path, unit, flags, min_chain):
    src = os.path.join(ROOT, "boom_amd", "csrc", unit + ".hip")
    asm = str(tmp_path / (unit + ".s"))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "-S",
                    "--cuda-device-only"] + flags + [src, "-o", asm], check=True, stderr=subprocess.DEVNULL,
                   cwd=os.path.dirname(src))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_serial_loads.py"), asm, str(min_chain)],
                         check=True, capture_output=True, text=True).stdout
    chains = [l for l in out.splitlines() if "global_load" in l or "buffer_load" in l or "flat_load" in l]
    if unit == "ss_round_kernel":   # (the capacity-16 instance BASELINE configs[2] runs; the larger ones still park registers)
        chains = [l for l in chains if "ss_round_kernelILi2E" in l]
    assert not chains, "\n".join(chains)
