import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from cases import bsts_priors, general_data, general_spec
from test_structural_general_gpu import make_engine
def mk(blocks, chains, T, p, kernel=None):
    seas = [(d[1], d[2]) for d in blocks if d[0] == "seasonal"]
    X, y, _, obs = general_data(T, p, 2, seas, seed=8, missing_frac=0.02, ar_coef=[0.5] if any(d[0]=="ar" for d in blocks) else None)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = general_spec(y, blocks)
    e = make_engine(chains, 7, y, X, obs, prior, spec, sig_up, np.zeros(p, np.uint8))
    if kernel is not None: e.ss_set_tuning(kernel=kernel)
    return e
for blocks, kernel in [([("trend",), ("seasonal", 7, 1)], None), ([("trend",), ("seasonal", 7, 1)], 0), ([("trend",), ("seasonal", 12, 1), ("ar", 1)], None), ([("level",), ("seasonal", 4, 3), ("ar", 2)], None)]:
    a = mk(blocks, 512, 200, 8, kernel); b = mk(blocks, 512, 200, 8, kernel)
    bad = 0
    for it in range(150):
        a.ss_sweep(1); b.ss_sweep(1)
        if it % 10 == 9:
            ga, ba_, sa = a.get_states(); gb, bb, sb = b.get_states()
            if not (np.array_equal(ga, gb) and np.array_equal(ba_, bb) and np.array_equal(sa, sb)):
                bad += 1
                d = np.where((ba_ != bb).any(1))[0]
                print(blocks, kernel, "it", it, "chains differing", d[:10], len(d)); break
            for c in (0, 100, 511):
                if not np.array_equal(a.ss_get_state_draw(c), b.ss_get_state_draw(c)):
                    print(blocks, kernel, "it", it, "state draw differs chain", c); bad += 1; break
            if bad: break
    print(blocks, "kernel", kernel, "deterministic" if not bad else "NOT deterministic")
