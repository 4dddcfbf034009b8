#!/usr/bin/env python3
"""The sufficient-statistics build (a1: xtx_mfma_kernel + plane_sum_kernel +
col_reduce_kernel) alone at one shape, design matrix drawn on the device, for the
profiler.  usage: suf_bench.py n p [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import boom_amd

n, p = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
gen = torch.Generator(device="cuda")
gen.manual_seed(8675309)
X = torch.randn((p, n), dtype=torch.float64, device="cuda", generator=gen)
y = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
torch.cuda.synchronize()
eng = boom_amd.Engine(4, seed=1)
eng.set_kernel_timing(True)
for _ in range(reps):
    eng.build_suf_from_xy_device(n, p, X.data_ptr(), y.data_ptr())
ms, cnt = eng.kernel_times()["xtx_mfma_kernel+plane_sum_kernel+col_reduce_kernel"]
print("suf build n=%d p=%d: %.3f ms per build (%d builds), %.2f TFLOP/s counted as n p^2"
      % (n, p, ms / cnt, cnt, n * float(p) * p / (ms / cnt * 1e-3) / 1e12))
