#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of ssvs_sweep_kernel from the -DBA_STAMPS
build (boom_amd/libboomamd_stamps.so).  Read the SHARES, not the run time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBN = os.environ.get("SUBSTAMPS", "0")
SUB = SUBN != "0"
os.environ["BOOM_AMD_LIB"] = os.path.join(ROOT, "tools", "build", {"0": "libboomamd_stamps.so", "1": "libboomamd_stamps2.so", "2": "libboomamd_stamps3.so", "3": "libboomamd_stamps4.so"}[SUBN])
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import time
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior

n, p = int(os.environ.get("N_ROWS", "10000")), int(os.environ.get("P_VARS", "512"))
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 16
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
X, y, _ = regression_data(n, p, nsig, seed=8675309)
eng = boom_amd.Engine(chains, seed=1, max_model_size_hint=int(os.environ.get('KCAP_HINT', '0')))
eng.set_tuning(waves_per_chain=int(os.environ.get("WAVES", "0")), walk_policy=int(os.environ.get("WALK", "-1")))
eng.build_suf_from_xy(X, y)
s = eng.get_suf()
suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"],
           sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
prior = spike_slab_prior(suf, nsig)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(p, np.uint8); g0[0] = 1
eng.set_state(g0)
eng.sweep(200)
eng.reset_summaries()
t0 = time.perf_counter()
NSW = int(os.environ.get("NSWEEP", "100"))
NL = int(os.environ.get("NLAUNCH", "1"))
for _ in range(NL):
    eng.sweep(NSW, sync=False)
eng.sync()
dt = time.perf_counter() - t0
sm = eng.get_summaries()
ph = sm["phase_cycles"]
names = ["shuffle uniforms", "shuffle serial", "refactor", "proposal batches",
         "swap", "sigma", "beta", "rest"]
if SUBN == "2":
    names = ["master: commit", "master: sweep start copy", "master: fork", "master: swap proposal",
             "master: sigma", "master: normals", "master: back substitution", "everything else"]
elif SUBN == "3":
    names = ["wave 1: shuffle uniforms", "wave 1: matching rounds", "wave 1: links", "wave 1: walks",
             "wave 1: table walk", "wave 1: waiting", "wave 1: proposal rounds", "wave 1: other"]
elif SUB:
    names = ["batch: index fetch", "batch: classify", "batch: V gather", "batch: V solve",
             "batch: A gather", "batch: A solve", "batch: epilogue", "outside batches"]
tot = ph.sum()
if not SUB:
    print("  per chain: mean %.0f cycles, slowest %.0f cycles (%.2fx)" % (tot / chains, sm["slot_hits"], sm["slot_hits"] * chains / tot))
print("waves=%s hint=%s" % (os.environ.get("WAVES","auto"), os.environ.get("KCAP_HINT","0")), end=" "); print("signals %d chains %d: %.1f us per sweep-round, kbar %.2f, accepts/sweep %.3f, proposals/sweep %.1f"
      % (nsig, chains, dt / (NSW * NL) * 1e6, sm["k_sum"] / sm["sweeps"], sm["accepts"] / sm["sweeps"],
         sm["proposals"] / sm["sweeps"]))
for nm, v in zip(names, ph):
    print("  %-26s %6.2f %%   %10.0f cycles/sweep" % (nm, 100 * v / tot, v / sm["sweeps"]))
