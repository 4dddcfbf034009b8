import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from cases import bsts_priors, structural_data, structural_spec
from oracle_lib import Oracle, ssvs_options
import test_structural_gpu as tg
o = Oracle()
trend, ns, T, coef = 1, 0, 150, [0.8]
p, chains, seed = 6, 5, 31
X, y, _, obs = structural_data(T, p, 2, ns, seed=3 + ns, ar_coef=coef)
prior, _, sig_up = bsts_priors(X, y, 2)
spec = structural_spec(y, trend, ns, ar_lags=len(coef))
opts = ssvs_options(sigma_upper_limit=sig_up)
g0 = np.zeros(p, np.uint8)
eng = tg.make_engine(chains, seed, y, X, obs, prior, spec, sig_up, g0)
r = o.ssm_run(y, X, obs, prior, opts, spec, ("philox", seed, 0), g0, 2)
eng.ss_impute_state()
ar = eng.ss_get_ar(0); st = eng.ss_get_structural(0)
print("after first impute: suf", ar["xtx"], ar["xty"], ar["yty"], ar["n"], "lev suf", st["suf_ss"])
blk = st["state"][:, 1:]
print("  from state:", blk[:-1].T @ blk[:-1], blk[:-1].T @ blk[1:, 0], (blk[1:, 0] ** 2).sum())
eng.ss_sweep(1)
ar = eng.ss_get_ar(0); st = eng.ss_get_structural(0)
print("dev phi", ar["phi"], "sig", ar["sigsq"], "var", st["variances"])
print("ora phi", r["ar_phi"][0], "sig", r["ar_sigsq"][0], "var", r["variances"][0])
print("state diff", np.abs(st["state"] - r["state"][0]).max(axis=0))
