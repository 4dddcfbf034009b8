import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import boom_amd
from cases import regression_data, spike_slab_prior, suf_from_xy
X, y, _ = regression_data(10000, 512, 16, seed=8675309)
suf = suf_from_xy(X, y)
prior = spike_slab_prior(suf, 16)
eng = boom_amd.Engine(1024, seed=1)
eng.build_suf_from_xy(X, y)
eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
g0 = np.zeros(512, np.uint8); g0[0] = 1
eng.set_state(g0); eng.sweep(1000)
L = 64
eng.set_lookahead(L)
for _ in range(L): eng.draw_next()
eng.get_state(0)
lib = eng.lib; h = eng._h
def timeit(fn, nb=20):
    tb = []
    for b in range(nb):
        t = time.perf_counter(); fn(); tb.append(time.perf_counter() - t)
    return np.median(tb)
def only_draw():
    for _ in range(L): lib.ba_draw_next(h)
    lib.ba_sync(h)
def draw_get():
    for _ in range(L):
        eng.draw_next(); eng.get_state(0)
def raw_draw_get():
    import ctypes as C
    g = np.zeros(512, np.uint8); b = np.zeros(512); s = C.c_double()
    gp = g.ctypes.data_as(C.POINTER(C.c_uint8)); bp = b.ctypes.data_as(C.POINTER(C.c_double))
    for _ in range(L):
        lib.ba_draw_next(h); lib.ba_get_state(h, 0, gp, bp, C.byref(s))
print("batch of %d: draw_next only + sync %.3f ms; raw ctypes draw+get %.3f ms; Engine wrappers %.3f ms" % (L, timeit(only_draw)*1e3, timeit(raw_draw_get)*1e3, timeit(draw_get)*1e3))
t0=time.perf_counter(); eng.set_lookahead(1); 
for _ in range(5): eng.sweep(64, sync=True)
t0=time.perf_counter()
for _ in range(10): eng.sweep(64, sync=True)
print("plain sweep(64)+sync: %.3f ms" % ((time.perf_counter()-t0)/10*1e3))
