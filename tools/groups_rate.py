import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, boom_amd
from cases import regression_data, spike_slab_prior
X, y, _ = regression_data(10000, 512, 16, seed=8675309)
def make(ch):
    e = boom_amd.Engine(ch, seed=1)
    e.build_suf_from_xy(X, y)
    s = e.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"]*s["n"], xsum=s["xbar"]*s["n"])
    pr = spike_slab_prior(suf, 16)
    e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
    g0 = np.zeros(512, np.uint8); g0[0] = 1
    e.set_state(g0); e.sweep(1000); e.sweep(1000)
    return e
for ch in (2048, 3072, 4096, 8192):
    for mode in ("groups", "one launch"):
        e = make(ch)
        if mode != "groups": e.set_kernel_timing(True)
        t0 = time.perf_counter()
        for _ in range(4): e.sweep(1000, sync=False)
        e.sync()
        dt = time.perf_counter() - t0
        print(ch, mode, "%.1f M sweeps/s" % (4 * ch * 1000 / dt / 1e6), flush=True)
        e.close()
