"""boom_amd -- MI355X-native many-chain engine for BOOM's spike-and-slab
(BregVsSampler) and bsts local-level (StateSpacePosteriorSampler) hot path.

The product is boom_amd/libboomamd.so (hand-written HIP for gfx950 behind the
C-ABI in include/boom_amd.h).  This package is the ctypes plumbing used by the
tests and bench.py; it never computes anything itself.
"""
from .capi import BoomAmdError, Engine, LIB_PATH, load_library  # noqa: F401
