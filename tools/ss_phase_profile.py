#!/usr/bin/env python3
"""Diagnostic: where the SSVS launch of a bsts round (config 3: one sweep per launch,
new X'y every time) spends its cycles -- the -DBA_STAMPS build's phase counters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBN = os.environ.get("SUBSTAMPS", "0")
os.environ["BOOM_AMD_LIB"] = os.path.join(ROOT, "tools", "build", {"0": "libboomamd_stamps.so", "1": "libboomamd_stamps2.so", "2": "libboomamd_stamps3.so", "3": "libboomamd_stamps4.so"}[SUBN])
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import boom_amd
from cases import bsts_priors, state_space_data
T, p, nsig, chains = 2000, 100, 5, 1024
X, y, btrue, _ = state_space_data(T, p, nsig, seed=8675309)
prior, ss, sig_up = bsts_priors(X, y, 5)
eng = boom_amd.Engine(chains, seed=4)
eng.ss_set_data(y, X, None)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"], sigma_upper_limit=sig_up)
eng.ss_set_local_level(ss["level_df"], ss["level_sigma_guess"], ss["level_sigma_upper_limit"],
                       ss["initial_state_mean"], ss["initial_state_variance"], ss["initial_level_sigma"])
eng.set_state(np.zeros(p, np.uint8))
eng.ss_sweep(50)
eng.reset_summaries()
eng.ss_sweep(100)
sm = eng.get_summaries()
ph = sm["phase_cycles"]
names = ["shuffle uniforms", "shuffle serial", "refactor", "proposal batches", "swap", "sigma", "beta", "rest"]
if SUBN == "1":
    names = ["batch: index fetch", "batch: classify", "batch: V gather", "batch: V solve",
             "batch: A gather", "batch: A solve", "batch: epilogue", "outside batches"]
elif SUBN == "2":
    names = ["master: commit", "master: sweep start copy", "master: fork", "master: swap proposal",
             "master: sigma", "master: normals", "master: back substitution", "everything else"]
elif SUBN == "3":
    names = ["wave 1: shuffle uniforms", "wave 1: matching rounds", "wave 1: links", "wave 1: walks",
             "wave 1: table walk", "wave 1: waiting", "wave 1: proposal rounds", "wave 1: other"]
tot = ph.sum()
print("per chain-sweep: %.0f cycles; slowest chain's launch total %.0f; kbar %.2f accepts/sweep %.3f"
      % (tot / sm["sweeps"], sm["slot_hits"], sm["k_sum"] / sm["sweeps"], sm["accepts"] / sm["sweeps"]))
for nm, v in zip(names, ph):
    print("  %-20s %6.2f %%   %8.0f cycles/sweep" % (nm, 100 * v / tot, v / sm["sweeps"]))
