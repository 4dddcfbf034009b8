#!/usr/bin/env python3
"""Random interleavings of ba_ss_draw_next / readers / mutators / plain sweeps / forecasts on
an engine that serves bsts's loop from look-ahead batches, against an engine that runs one
round per call: everything compared must be equal, bit for bit (diagnostic; the same checks
at fixed small sizes live in tests/test_ss_lookahead_gpu.py).
usage: ss_la_stress.py [iterations per model [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import bsts_priors, general_data, general_spec, state_space_data
from test_state_space_gpu import make_engine as level_engine
from test_structural_general_gpu import make_engine as general_engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.Generator(np.random.PCG64(seed))


def same(a, b, what):
    if isinstance(a, dict):
        for k in a:
            same(a[k], b[k], what + "." + k)
    elif isinstance(a, (tuple, list)):
        for i, (u, v) in enumerate(zip(a, b)):
            same(u, v, what + "[%d]" % i)
    else:
        assert np.array_equal(np.asarray(a), np.asarray(b)), what


def run(name, make, blocks, chains, T, p, L):
    if blocks is None:
        X, y, _, obs = state_space_data(T, p, 3, seed=seed + 5, missing_frac=0.02)
        prior, ss, sig_up = bsts_priors(X, y, 3)
        a = make(chains, 7, y, X, obs, prior, ss, sig_up, np.zeros(p, np.uint8))
        b = make(chains, 7, y, X, obs, prior, ss, sig_up, np.zeros(p, np.uint8))
    else:
        seas = [(d[1], d[2]) for d in blocks if d[0] == "seasonal"]
        X, y, _, obs = general_data(T, p, 2, seas, seed=seed + 6, missing_frac=0.02,
                                    ar_coef=[0.5] if any(d[0] == "ar" for d in blocks) else None)
        prior, _, sig_up = bsts_priors(X, y, 2)
        spec = general_spec(y, blocks)
        a = make(chains, 7, y, X, obs, prior, spec, sig_up, np.zeros(p, np.uint8))
        b = make(chains, 7, y, X, obs, prior, spec, sig_up, np.zeros(p, np.uint8))
    if blocks is not None and os.environ.get("SS_KERNEL"):
        a.ss_set_tuning(kernel=int(os.environ["SS_KERNEL"]))
        b.ss_set_tuning(kernel=int(os.environ["SS_KERNEL"]))
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
    print("%-28s %d iterations, look-ahead %d, %d chains: equal throughout (%.1f s) %s"
          % (name, iters, L, chains, time.perf_counter() - t0, counts), flush=True)
    a.close(); b.close()


only = os.environ.get("SS_ONLY", "")
if not only:
    run("local level", level_engine, None, 48, 300, 12, int(rng.integers(3, 20)))
run("trend + 7 seasons (template)", general_engine, [("trend",), ("seasonal", 7, 1)], 40, 200, 8, int(rng.integers(3, 20)))
run("level + seasonal(4x3) + ar(2)", general_engine, [("level",), ("seasonal", 4, 3), ("ar", 2)], 33, 150, 8, int(rng.integers(3, 20)))
run("trend + 12 seasons + ar(1)", general_engine, [("trend",), ("seasonal", 12, 1), ("ar", 1)], 24, 130, 6, int(rng.integers(3, 20)))
