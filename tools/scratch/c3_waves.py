import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import boom_amd
from cases import bsts_priors, state_space_data
T, p = 2000, 100
X, y, _, _ = state_space_data(T, p, 5, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
for chains, waves, kstart in ((512, 1, 0), (512, 2, 0), (2048, 1, 0), (2048, 2, 0), (4096, 1, 0), (4096, 2, 0), (1024, 1, 0), (1024, 2, 0)):
    eng = boom_amd.Engine(chains, seed=4)
    eng.set_tuning(waves, -1, kstart)
    eng.ss_set_data(y, X, None)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                           ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_sweep(50)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.ss_sweep(200); ts.append((time.perf_counter() - t0) / 200)
    eng.set_kernel_timing(True)
    eng.ss_sweep(100)
    kt = eng.kernel_times()
    print("chains %d waves %d: %.1f us per round" % (chains, waves, np.median(ts) * 1e6),
          {k: round(v[0] / v[1] * 1e3, 1) for k, v in kt.items()}, flush=True)
    eng.close()
