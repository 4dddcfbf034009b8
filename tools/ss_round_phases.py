#!/usr/bin/env python3
"""Where a chain's round goes in the persistent round kernel (ss_round_kernel.hip): wall-clock
microseconds per phase and wave, mean / p95 / max over the chains, from the diagnostic build
  make -C boom_amd/csrc ../../tools/build/libboomamd_rstamps.so
(BOOM_AMD_LIB names it).  BASELINE configs[2]: T=2000 p=100, 1024 chains."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BOOM_AMD_LIB", os.path.join(ROOT, "tools", "build", "libboomamd_rstamps.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, state_space_data
T, p, nsig = 2000, 100, 5
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X, y, btrue, _ = state_space_data(T, p, nsig, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
eng = boom_amd.Engine(chains, seed=4)
eng.ss_set_data(y, X, None)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                       ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
eng.set_state(np.zeros(p, np.uint8))
eng.ss_sweep(200)
lib = eng.lib
lib.ba_debug_round_stamps.restype = C.c_int
lib.ba_debug_round_stamps.argtypes = [C.c_void_p, C.c_int64]
buf = np.zeros(chains * 16)
lib.ba_debug_round_stamps(buf.ctypes.data, buf.size)   # (reset)
import time
t0 = time.perf_counter()
try:
    eng.ss_sweep(rounds)
except Exception as ex:
    print("FAILED:", ex)
print("host wall: %.1f us per round" % ((time.perf_counter() - t0) / rounds * 1e6))
n = lib.ba_debug_round_stamps(buf.ctypes.data, buf.size)
assert n == buf.size, n
us = buf.reshape(chains, 2, 8) / 100.0 / rounds
names = ["sweep | variance + normals", "waiting for the other wave", "state draw", "joining / tile closes",
         "share of the product", "waiting for the members", "plane sum + record"]
for w in (0, 1):
    print("wave %d: us per round  mean   p95    max" % w)
    for i, nm in enumerate(names):
        v = us[:, w, i]
        print("  %-30s %6.1f %6.1f %6.1f" % (nm, v.mean(), np.percentile(v, 95), v.max()))
    print("  %-30s %6.1f" % ("sum", us[:, w, :7].sum(1).mean()))
    print("  diagnostic counts (stale series + 1e6 x NaN sums): %d in chains %s" % (
        int((us[:, w, 7] * 100 * rounds).sum()), np.nonzero(us[:, w, 7])[0][:10]))

if rounds <= 64:
    raw = buf.reshape(chains, 2, 8)
    st, en = raw[:, 1, 5], raw[:, 1, 6]
    t0 = st.min()
    print("workgroup start / end (us after the first start), by quarter of the grid:")
    for q in range(4):
        sl = slice(q * chains // 4, (q + 1) * chains // 4)
        print("  chains %4d-%4d: start %7.1f .. %7.1f   end %8.1f .. %8.1f" % (
            sl.start, sl.stop - 1, (st[sl].min() - t0) / 100, (st[sl].max() - t0) / 100,
            (en[sl].min() - t0) / 100, (en[sl].max() - t0) / 100))
