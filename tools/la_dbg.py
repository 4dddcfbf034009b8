import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from cases import bsts_priors, general_data, general_spec
from test_structural_general_gpu import make_engine
blocks = [("trend",), ("seasonal", 7, 1)]
T, p, chains = 200, 8, 40
X, y, _, obs = general_data(T, p, 2, [(7, 1)], seed=8, missing_frac=0.02)
prior, _, sig_up = bsts_priors(X, y, 2)
spec = general_spec(y, blocks)
def mk():
    e = make_engine(chains, 7, y, X, obs, prior, spec, sig_up, np.zeros(p, np.uint8))
    if os.environ.get("KCAP"): e.set_tuning(kcap_start=int(os.environ["KCAP"]))
    if os.environ.get("SS_KERNEL"): e.ss_set_tuning(kernel=int(os.environ["SS_KERNEL"]))
    return e
def eq(a, b):
    return all(np.array_equal(u, v) for u, v in zip(a.get_states(), b.get_states()))
for scenario in ["none", "forecast", "set_options", "set_state", "plain", "unrecorded"]:
    a, b = mk(), mk()
    b.ss_set_lookahead(6)
    newX = np.random.Generator(np.random.PCG64(1)).standard_normal((4, p))
    ok = True
    for it in range(40):
        a.ss_sweep(1); b.ss_draw_next()
        if not eq(a, b): print(scenario, "differs at it", it, "(before the event)"); ok = False; break
        if it % 5 == 3:
            if scenario == "forecast":
                fa, fb = a.ss_forecast(newX), b.ss_forecast(newX)
                if not np.array_equal(fa, fb): print(scenario, "forecast differs at it", it, np.abs(fa-fb).max()); ok = False; break
            elif scenario == "set_options":
                a.set_options(max_flips=3 + it % 4); b.set_options(max_flips=3 + it % 4)
            elif scenario == "set_state":
                g = np.zeros(p, np.uint8); g[it % p] = 1
                a.set_state(g, chain=5); b.set_state(g, chain=5)
            elif scenario == "plain":
                a.ss_sweep(2); b.ss_sweep(2)
            elif scenario == "unrecorded":
                if not np.array_equal(a.ss_get_state_draw(17), b.ss_get_state_draw(17)): print(scenario, "state draw differs", it); ok = False; break
    print(scenario, "ok" if ok else "FAILED")
