import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import boom_amd
from cases import bsts_priors, structural_data, structural_spec
from oracle_lib import Oracle, ssvs_options
from test_structural_gpu import make_engine
O = Oracle()
for (trend, ns, T) in [(2, 12, 300), (2, 12, 100), (1, 12, 100), (2, 8, 100), (2, 9, 100), (2, 10, 100), (2, 11, 100), (1, 9, 100), (1,10,100)]:
    p, chains, seed, nsw = 6, 5, 29, 1
    X, y, _, obs = structural_data(T, p, 2, ns, seed=3 + ns)
    prior, _, sig_up = bsts_priors(X, y, 2)
    spec = structural_spec(y, trend, ns)
    opts = ssvs_options(sigma_upper_limit=sig_up)
    g0 = np.zeros(p, np.uint8)
    eng = make_engine(chains, seed, y, X, obs, prior, spec, sig_up, g0)
    o = O.ssm_run(y, X, obs, prior, opts, spec, ("philox", seed, 0), g0, nsw)
    for s in range(nsw):
        try:
            eng.ss_sweep(1)
        except Exception as e:
            print("sweep failed", e); break
        st = eng.ss_get_structural(0)
        d = np.abs(st["state"] - o["state"][s])
        t_bad = np.where(d.max(axis=1) > 1e-8)[0]
        print(trend, ns, T, "sweep", s, "max state diff %.3g" % d.max(), "first bad t", t_bad[:3], "var", st["variances"], o["variances"][s])
