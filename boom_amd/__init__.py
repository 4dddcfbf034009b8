"""boom_amd -- MI355X-native many-chain engine for BOOM's spike-and-slab
(BregVsSampler) and bsts local-level (StateSpacePosteriorSampler) hot path.

The product is boom_amd/libboomamd.so (hand-written HIP for gfx950 behind the
C-ABI in include/boom_amd.h).  This package is the ctypes plumbing used by the
tests and bench.py; it never computes anything itself.
"""
# PyTorch ships its own copy of the HIP runtime under the same soname as
# /opt/rocm's: a process gets whichever is loaded first, and torch.cuda cannot
# initialise on top of the system copy ("No HIP GPUs are available").  Import
# torch before the library is mapped so that both use one runtime.
try:
    import torch  # noqa: F401
except ImportError:   # (the C-ABI library itself does not need PyTorch)
    pass

from .capi import BoomAmdError, Engine, Group, LIB_PATH, load_library  # noqa: F401
