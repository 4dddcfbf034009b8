#!/usr/bin/env python3
"""BASELINE configs[2] (T=2000, p=100, 1024 chains): the persistent round kernel against the
separate launches per round, every chain compared after each call of `step` rounds (argv[1]),
40 calls.  DEBUG=1 in the environment (-> ba_ss_set_tuning(e, 6)) prints what the kernel noted when a chain stops."""
import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, state_space_data
T, p, nsig, chains = 2000, 100, 5, 1024
X, y, btrue, _ = state_space_data(T, p, nsig, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
def mk(kernel):
    eng = boom_amd.Engine(chains, seed=4)
    eng.ss_set_data(y, X, None)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
    eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                           ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
    eng.set_state(np.zeros(p, np.uint8))
    eng.ss_set_tuning(kernel=kernel)
    if kernel == 5 and os.environ.get("DEBUG"):
        eng.ss_set_tuning(kernel=6)
    return eng
a, b = mk(5), mk(4)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for it in range(40):
    try:
        a.ss_sweep(step)
    except Exception as ex:
        print("round kernel failed at iteration", it, ex); break
    b.ss_sweep(step)
    ga, ba, sa = a.get_states(); gb, bb, sb = b.get_states()
    bad = np.where((ga != gb).any(1) | (np.abs(sa - sb) > 1e-8 * sb))[0]
    if len(bad):
        print("iteration", it, "rounds", (it + 1) * step, "chains differing:", bad[:20], len(bad))
        c = bad[0]
        print(" sig", sa[c], sb[c], "gamma diff", np.where(ga[c] != gb[c])[0])
        xa, xb = a.ss_get_chain_suf(c), b.ss_get_chain_suf(c)
        print(" xty max rel diff", np.max(np.abs(xa["xty"] - xb["xty"])) / np.abs(xb["xty"]).max(), xa["yty"], xb["yty"])
        break
else:
    print("no difference in", 40 * step, "rounds")
try:
    a.ss_sweep(1)   # (prints the round kernel's diagnostics of the failed call, if DEBUG)
except Exception as ex:
    pass
