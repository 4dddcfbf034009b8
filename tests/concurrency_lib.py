"""Concurrency checks shared by tests/test_concurrency_gpu.py (bounded) and the long-running
tools (tools/ss_la_stress.py, tools/concurrency_check.py): random interleavings of the bsts
look-ahead against a one-round-per-call engine, and every sampler family alone against the
same family beside a second engine that keeps the GPU busy.  Equality is bitwise: what else
shares the machine may change the timing inside a kernel, never a draw."""
import time

import numpy as np

from cases import (bsts_priors, general_data, general_spec, logit_data, probit_data, probit_slab,
                   regression_data, spike_slab_prior, state_space_data)


def same(a, b, what):
    if isinstance(a, dict):
        for k in a:
            same(a[k], b[k], what + "." + k)
    elif isinstance(a, (tuple, list)):
        for i, (u, v) in enumerate(zip(a, b)):
            same(u, v, what + "[%d]" % i)
    else:
        assert np.array_equal(np.asarray(a), np.asarray(b)), what


LA_MODELS = {
    "local level": (None, 48, 300, 12),
    "trend + 7 seasons (template)": ([("trend",), ("seasonal", 7, 1)], 40, 200, 8),
    "level + seasonal(4x3) + ar(2)": ([("level",), ("seasonal", 4, 3), ("ar", 2)], 33, 150, 8),
    "trend + 12 seasons + ar(1)": ([("trend",), ("seasonal", 12, 1), ("ar", 1)], 24, 130, 6),
}


def la_stress(name, iters, seed, kernel=None, verbose=False):
    """ba_ss_draw_next + readers / mutators / plain sweeps / forecasts in random order on an
    engine behind the look-ahead, against one round per call: equal at every step"""
    from test_state_space_gpu import make_engine as level_engine
    from test_structural_general_gpu import make_engine as general_engine
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks, chains, T, p = LA_MODELS[name]
    L = int(rng.integers(3, 20))
    if blocks is None:
        X, y, _, obs = state_space_data(T, p, 3, seed=seed + 5, missing_frac=0.02)
        prior, ss, sig_up = bsts_priors(X, y, 3)
        a = level_engine(chains, 7, y, X, obs, prior, ss, sig_up, np.zeros(p, np.uint8))
        b = level_engine(chains, 7, y, X, obs, prior, ss, sig_up, np.zeros(p, np.uint8))
    else:
        seas = [(d[1], d[2]) for d in blocks if d[0] == "seasonal"]
        X, y, _, obs = general_data(T, p, 2, seas, seed=seed + 6, missing_frac=0.02,
                                    ar_coef=[0.5] if any(d[0] == "ar" for d in blocks) else None)
        prior, _, sig_up = bsts_priors(X, y, 2)
        spec = general_spec(y, blocks)
        a = general_engine(chains, 7, y, X, obs, prior, spec, sig_up, np.zeros(p, np.uint8))
        b = general_engine(chains, 7, y, X, obs, prior, spec, sig_up, np.zeros(p, np.uint8))
        if kernel is not None:
            a.ss_set_tuning(kernel=kernel)
            b.ss_set_tuning(kernel=kernel)
    watch = sorted(set(int(c) for c in rng.integers(0, chains, 2)) | {0})
    b.ss_set_lookahead(L, chains=watch)
    newX = rng.standard_normal((4, p))
    t0 = time.perf_counter()
    counts = {}
    for it in range(iters):
        a.ss_sweep(1)
        b.ss_draw_next()
        c = int(rng.choice(watch))
        same(a.get_state(c), b.get_state(c), "%s it %d get_state(%d)" % (name, it, c))
        if blocks is None:
            same(a.ss_get_state(c, suf=False), b.ss_get_state(c, suf=False), "%s it %d state(%d)" % (name, it, c))
        else:
            same(a.ss_get_state_draw(c), b.ss_get_state_draw(c), "%s it %d draw(%d)" % (name, it, c))
        u = rng.random()
        ev = None
        if u < 0.04:
            ev = "all states"
            same(a.get_states(), b.get_states(), "%s it %d get_states" % (name, it))
        elif u < 0.07:
            ev = "unrecorded chain"
            cc = int(rng.integers(0, chains))
            if blocks is None:
                same(a.ss_get_state(cc), b.ss_get_state(cc), "%s it %d full state(%d)" % (name, it, cc))
            else:
                k = int(rng.integers(0, len(blocks)))
                same(a.ss_get_state_model(cc, k), b.ss_get_state_model(cc, k), "%s it %d model(%d,%d)" % (name, it, cc, k))
                same(a.ss_get_state_draw(cc), b.ss_get_state_draw(cc), "%s it %d draw(%d)" % (name, it, cc))
        elif u < 0.10:
            ev = "mutator"
            mf = int(rng.integers(2, p + 1))
            a.set_options(max_flips=mf)
            b.set_options(max_flips=mf)
        elif u < 0.12:
            ev = "set_state"
            g = (rng.random(p) < 0.2).astype(np.uint8)
            cc = int(rng.integers(0, chains))
            a.set_state(g, chain=cc)
            b.set_state(g, chain=cc)
        elif u < 0.14:
            ev = "plain sweeps"
            k = int(rng.integers(1, 4))
            a.ss_sweep(k)
            b.ss_sweep(k)
        elif u < 0.16:
            ev = "forecast"
            same(a.ss_forecast(newX), b.ss_forecast(newX), "%s it %d forecast" % (name, it))
        if ev:
            counts[ev] = counts.get(ev, 0) + 1
    same(a.get_states(), b.get_states(), name + " end")
    if verbose:
        print("%-28s %d iterations, look-ahead %d, %d chains: equal throughout (%.1f s) %s"
              % (name, iters, L, chains, time.perf_counter() - t0, counts), flush=True)
    a.close()
    b.close()


# ---- every family alone and beside a busy second engine ---------------------------------
def noise_engine():
    import boom_amd
    X, y, _ = regression_data(4000, 256, 12, seed=3)
    e = boom_amd.Engine(1024, seed=99)
    e.build_suf_from_xy(X, y)
    s = e.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    pr = spike_slab_prior(suf, 12)
    e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
    g0 = np.zeros(256, np.uint8)
    g0[0] = 1
    e.set_state(g0)
    e.sweep(50)
    return e


def families():
    """name -> (make engine, one asynchronous step, extra arrays to compare)"""
    import boom_amd
    from test_state_space_gpu import make_engine as level_engine
    from test_structural_general_gpu import make_engine as general_engine
    fam = {}
    X, y, _, obs = state_space_data(500, 20, 3, seed=5, missing_frac=0.02)
    prior, ss, sig_up = bsts_priors(X, y, 3)
    fam["bsts local level"] = (lambda: level_engine(256, 7, y, X, obs, prior, ss, sig_up, np.zeros(20, np.uint8)),
                               lambda e, s: e.ss_sweep(5, sync=s),
                               lambda e: [e.ss_get_state(3)["state"], e.ss_get_state(200)["state"]])
    for nm, desc in [("structural template trend+12", [("trend",), ("seasonal", 12, 1)]),
                     ("structural template +ar(2)", [("trend",), ("seasonal", 7, 1), ("ar", 2)]),
                     ("structural general 4x3 + ar", [("level",), ("seasonal", 4, 3), ("ar", 2)]),
                     ("structural general m=27", [("trend",), ("seasonal", 7, 1), ("seasonal", 20, 2)])]:
        seas = [(d[1], d[2]) for d in desc if d[0] == "seasonal"]
        Xg, yg, _, og = general_data(300, 8, 2, seas, seed=8, missing_frac=0.02,
                                     ar_coef=[0.5] if any(d[0] == "ar" for d in desc) else None)
        pg, _, su = bsts_priors(Xg, yg, 2)
        spec = general_spec(yg, desc)
        fam[nm] = (lambda Xg=Xg, yg=yg, og=og, pg=pg, spec=spec, su=su:
                   general_engine(200, 7, yg, Xg, og, pg, spec, su, np.zeros(8, np.uint8)),
                   lambda e, s: e.ss_sweep(4, sync=s),
                   lambda e: [e.ss_get_state_draw(0), e.ss_get_state_draw(150)])
    Xr, yr, _ = regression_data(5000, 300, 10, seed=4)

    def reg():
        e = boom_amd.Engine(512, seed=5)
        e.build_suf_from_xy(Xr, yr)
        s = e.get_suf()
        suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
        pr = spike_slab_prior(suf, 10)
        e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
        g0 = np.zeros(300, np.uint8)
        g0[0] = 1
        e.set_state(g0)
        return e
    fam["BregVsSampler sweeps"] = (reg, lambda e, s: e.sweep(40, sync=s), None)
    # models of 33 .. 128 variables: the table fills on the matrix cores (ssvs_fill_mfma.h) read
    # the model block another wavefront published -- at capacity 48 / 64 in the LDS kernel,
    # and in the large-model kernel
    for nm, nsig in [("BregVsSampler, 40 signals", 40), ("BregVsSampler, 70 signals", 70)]:
        Xd, yd, _ = regression_data(3000, 200, nsig, seed=40 + nsig)

        def dense(Xd=Xd, yd=yd, nsig=nsig):
            e = boom_amd.Engine(128, seed=9)
            e.build_suf_from_xy(Xd, yd)
            s = e.get_suf()
            suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
            pr = spike_slab_prior(suf, nsig)
            e.set_priors(pr["b"], pr["ominv"], pr["pi"], pr["df"], pr["sigma_guess"])
            g0 = np.zeros(200, np.uint8)
            g0[0] = 1
            e.set_state(g0)
            e.sweep(60)   # (past the growth through the capacities)
            return e
        fam[nm] = (dense, lambda e, s: e.sweep(15, sync=s), None)
    fam["adaptive sampler"] = (reg, lambda e, s: e.adaptive_sweep(20, sync=s), None)
    for kind, data in [("probit", probit_data), ("logit", logit_data)]:
        Xl, yl, nt, _ = data(3000, 40, 4, seed=6)
        slab, pi = probit_slab(Xl, nt, 4)

        def glm(kind=kind, Xl=Xl, yl=yl, nt=nt, slab=slab, pi=pi):
            e = boom_amd.Engine(128, seed=9)
            (e.probit_set_data if kind == "probit" else e.logit_set_data)(Xl, yl, nt, 5)
            e.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
            e.set_spike(pi)
            g0 = np.zeros(40, np.uint8)
            g0[0] = 1
            e.set_state(g0)
            return e
        fam[kind + " spike-and-slab"] = (glm, (lambda e, s, kind=kind: (e.probit_sweep if kind == "probit" else e.logit_sweep)(3, sync=s)), None)
    return fam


def alone_vs_loaded(make, step, extra, steps=6, noise=None):
    """the number of compared arrays that differ between a run alone and a run beside the
    noise engine's asynchronous launches"""
    def states(e):
        return [np.array(x) for x in e.get_states()]
    a = make()
    for _ in range(steps):
        step(a, True)
    ra = states(a) + (extra(a) if extra else [])
    b = make()
    nz = noise if noise is not None else noise_engine()
    for _ in range(steps):
        nz.sweep(30, sync=False)      # the noise runs on its own stream beside what follows
        step(b, False)
    b.sync()
    nz.sync()
    rb = states(b) + (extra(b) if extra else [])
    bad = sum(0 if np.array_equal(u, v) else 1 for u, v in zip(ra, rb))
    a.close()
    b.close()
    if noise is None:
        nz.close()
    return bad
