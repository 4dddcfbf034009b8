#!/usr/bin/env python3
"""Headline benchmark: Gibbs sweeps/sec (all chains) for BOOM's spike-and-slab
sampler at BASELINE.json config 2 -- n=1e4, p=512, 1024 chains per MI355X, fp64.

  python bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no torch.distributed environment the script starts its own
N ranks (python -m torch.distributed.run ... bench.py) before anything touches a
GPU and exits with that job's code; under torch.distributed.run it is a rank.

A "step" is one pass of the hot path over one batch: SWEEPS_PER_STEP
BregVsSampler::draw() sweeps of every chain resident on the GPU, issued as one
kernel launch through the C-ABI.  Inputs (XtX, priors, chain state) are resident
in HBM before the timed region.  Weak scaling: every rank owns 1024 chains
(global chain ids rank*1024 ..), no collective on the data path, one RCCL
all-gather of the posterior-summary block at the end.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_OBS, P, N_SIGNAL = 10000, 512, 16
CHAINS_PER_GPU = 1024
SWEEPS_PER_STEP = 1000  # "200 burn-in + 1000 timed sweeps" -- one launch
BURN_IN = 1000   # (every launch of the run is 1000 sweeps, so that a profiler's per-kernel average is the step time)
ESS_SWEEPS = 1000
DATA_SEED = 8675309
SAMPLER_SEED = 8675309
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def geyer_ess(x):
    """ESS of one scalar trace by Geyer's initial positive sequence on the
    autocorrelation (BOOM has no ESS routine; stats/acf.hpp:28 is its only
    building block -- SURVEY 8d)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    x = x - x.mean()
    var = float(x @ x) / n
    if var <= 0 or n < 4:
        return float(n)
    f = np.fft.rfft(x, 2 * n)
    acf = np.fft.irfft(f * np.conj(f))[:n].real / (n * var)
    s = 0.0
    for m in range(0, n // 2 - 1):
        pair = acf[2 * m] + acf[2 * m + 1]
        if pair <= 0:
            break
        s += pair
    tau = max(2.0 * s - 1.0, 1.0 / n)
    return min(float(n), n / tau)


def usable_cores():
    """host threads this process may actually use: the affinity mask capped by a
    cgroup CPU quota (os.cpu_count() reports the machine, not the container)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            with open(path) as fh:
                txt = fh.read().strip()
            if parse:
                q, per = parse(txt)
                if q != "max":
                    n = min(n, max(1, int(float(q) / float(per) + 0.5)))
            else:
                q = int(txt)
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    per = int(fh.read().strip())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


F64_MATRIX_PEAK_TF = 78.6   # MI355X FP64 matrix spec (the guide's table stops at FP32; AMD data sheet)



# ---- executed-work roofline of a latency / issue bound kernel (VERDICT r5 task 1) ----------------
# SURVEY 8(d)'s byte formulas count what the REFERENCE's algorithm would read per unit; the
# kernels here do not do that work (threshold tables, factors kept in LDS, scans), and most are
# bound by how fast a few wavefronts can issue dependent instructions, not by a pipe's width.
# For those the fraction that means something is: of a wavefront's lifetime, how much is
# accounted for by the instructions it executed at the guide's issue cost (one wave issues one
# instruction per 4 cycles: MI355X_MICROARCH.md, "vector-instruction ISSUE cost", and `s_nop 0`
# 4; the SQ counters count in units of 4 cycles, so N instructions = N counter units) plus the
# dependent memory round trips its algorithm cannot overlap with anything (itemised by the
# caller: LDS ~64 cycles, L2 hit ~200: same guide).  Inputs: the newest committed counter passes
# profiles/r*_<tag>_pmc_summary.json (tools/regen_profiles.sh; separate --pmc passes of the same
# command, one launch at a time), named in the output with their commit.
ISSUE_CYCLES, LDS_TRIP_CYCLES, L2_TRIP_CYCLES = 4.0, 64.0, 200.0


def _executed(tag, kernel_prefix, units_per_dispatch=None, dependent_trips=None):
    """per-wave executed work of the kernel's most-dispatched instance, or None.  `units_per_dispatch`
    (sweeps, rounds ... of ALL chains a dispatch runs) scales the per-unit figures; `dependent_trips`
    = {"lds": n, "l2": n, "what": str} per unit of one chain's critical wavefront."""
    try:
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_pmc_summary.json" % tag)))
        with open(cand[-1]) as fh:
            pj = json.load(fh)
        ent = [(k, v) for k, v in pj["kernels"].items() if k.startswith(kernel_prefix)]
        name, v = max(ent, key=lambda kv: kv[1]["counters_avg_per_dispatch"]["SQ_WAVE_CYCLES"]["mean"]
                      * kv[1]["counters_avg_per_dispatch"]["SQ_WAVE_CYCLES"]["dispatches"])
        c = {k: x["mean"] for k, x in v["counters_avg_per_dispatch"].items()}
        waves = c["SQ_WAVES"]
        insts = {k: c.get("SQ_INSTS_" + k, 0.0) for k in ("VALU", "SALU", "LDS", "SMEM", "VMEM_RD", "VMEM_WR")}
        n_inst = sum(insts.values())
        wave_cycles = c["SQ_WAVE_CYCLES"] * 4.0 / waves                 # cycles of one wave's lifetime
        issue = n_inst * ISSUE_CYCLES / waves                           # ... its instructions at the issue cost
        out = {"kernel": name, "source": "profiles/%s at commit %s" % (os.path.basename(cand[-1]), pj.get("commit")),
               "launch": {k: v["launch_config"].get(k) for k in ("workgroup", "vgpr", "sgpr", "scratch", "lds")},
               "waves_per_dispatch": waves,
               "wave_instructions_per_dispatch": {k.lower(): round(x, 0) for k, x in insts.items()},
               "lds_cycles_per_dispatch": round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) * 4.0, 0),
               "lds_bank_conflict_frac": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
               "busy_cycles_per_dispatch": round(c.get("SQ_BUSY_CYCLES", 0.0) * 4.0, 0),
               "issue_occupancy": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
               "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
               "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
               "cycles_per_wave": round(wave_cycles, 0),
               "issue_bound_cycles_per_wave": round(issue, 0)}
        out["_n_inst"], out["_waves"], out["_wave_cycles_total"] = n_inst, waves, c["SQ_WAVE_CYCLES"] * 4.0
        if units_per_dispatch:
            # a unit (a sweep, a round) of ONE chain: the chain's waves_per_chain wavefronts each
            # live cycles_of_each_wave through it
            out["per_unit_of_one_chain"] = {
                "units_per_dispatch": units_per_dispatch,
                "wave_instructions": {k.lower(): round(x / units_per_dispatch, 1) for k, x in insts.items()},
                "lds_cycles": round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) * 4.0 / units_per_dispatch, 1)}
        if dependent_trips:
            out["dependent_trips"] = dependent_trips
        return out
    except Exception:
        return None


def _latency_issue_roofline(ex, rate, unit, chains, units_per_chain_per_dispatch, waves_per_chain, reference_bytes):
    """the contract's roofline object for a kernel bound by dependent-instruction latency and
    issue: `achieved` = the measured rate, `peak` = the rate at which every wavefront would
    spend its whole lifetime issuing the instructions it executed (4 cycles each) and waiting
    for the round trips its algorithm serialises -- i.e. achieved / frac --, `frac` = that
    lower bound in cycles over the measured cycles, both from the counter passes `ex` names.
    The SURVEY 8(d) byte figure stays, under the name of what it is."""
    out = {"bound": "latency/issue", "achieved": round(rate, 1), "peak": None, "unit": unit, "frac": None,
           "traffic": None, "work_rate_vs_reference_bytes": reference_bytes, "executed": None}
    if ex is None:
        out["note"] = "no counter passes committed for this kernel: frac not derived"
        return out
    per_unit_cycles = ex["_wave_cycles_total"] / ex["_waves"] / units_per_chain_per_dispatch   # of each of the chain's waves
    issue = ex["_n_inst"] * ISSUE_CYCLES / ex["_waves"] / units_per_chain_per_dispatch
    trips = ex.get("dependent_trips") or {}
    lat = trips.get("lds", 0) * LDS_TRIP_CYCLES + trips.get("l2", 0) * L2_TRIP_CYCLES
    bound = issue + lat
    frac = bound / per_unit_cycles
    exo = {k: v for k, v in ex.items() if not k.startswith("_")}
    exo["measured_cycles_per_unit_per_wave"] = round(per_unit_cycles, 0)
    exo["bound_cycles_per_unit_per_wave"] = {"issue": round(issue, 0), "dependent_round_trips": round(lat, 0),
                                             "total": round(bound, 0),
                                             "how": "wave-instructions of the chain's %d wavefront(s) / %d x %g cycles "
                                                    "+ LDS trips x %g + L2 trips x %g (MI355X_MICROARCH.md cycle "
                                                    "constants)" % (waves_per_chain, waves_per_chain, ISSUE_CYCLES,
                                                                    LDS_TRIP_CYCLES, L2_TRIP_CYCLES)}
    out.update({"peak": round(rate / frac, 1), "frac": round(frac, 4), "executed": exo})
    return out


def _rebase(old, tag, prefix, rate, unit, waves_per_chain):
    """a side configuration's roofline whose own note says "not bandwidth": the same executed-work
    form as the headline's (bound latency/issue; frac from the committed counter passes of the
    workload `tag`), SURVEY 8(d)'s byte rate kept under work_rate_vs_reference_bytes"""
    ref = {k: v for k, v in old.items() if k not in ("bound", "traffic", "traffic_source", "kernel", "note")}
    ref["what"] = "SURVEY 8(d) / DESIGN 3.x algorithmic bytes over the kernel's time, against the HBM peak: a work rate"
    new = _latency_issue_roofline(_executed(tag, prefix), rate, unit, None, 1, waves_per_chain, ref)
    for k in ("kernel", "traffic", "traffic_source"):
        if k in old:
            new[k] = old[k]
    if "note" in old:
        new["note"] = old["note"] + "; frac = executed instructions at the guide's 4-cycle issue cost over the wavefronts' measured lifetime"
    return new


def _per_launch(times):
    return {k: round(ms / max(1, n) * 1e3, 2) for k, (ms, n) in times.items()}   # microseconds


def _cpu_rate(fn, nchains, cores, target_s=6.0, first=4):
    """sweeps/s of `fn(chain, nsweeps)` run for nchains chains on `cores` threads:
    one short calibration run, then a run sized for about target_s seconds"""
    from concurrent.futures import ThreadPoolExecutor

    def timed(nsw):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda c: fn(c, nsw), range(nchains)))
        return time.perf_counter() - t0
    dt = timed(first)
    nsw = int(max(first, min(2000, target_s / max(dt / first, 1e-9))))
    if nsw > first:
        dt = timed(nsw)
    else:
        nsw = first
    return nchains * nsw / dt, nsw


def other_configs(boom_amd, torch, device, cpu=True):
    from cases import bsts_priors, logit_data, probit_slab, spike_slab_prior, state_space_data
    from oracle_lib import Oracle, ssvs_options
    cores = usable_cores()
    O = Oracle() if cpu else None
    other = {}

    # ---- configs[2]: bsts local level + regression, T=2000 p=100, 1024 chains ---------
    T3, p3, C3 = 2000, 100, 1024
    Xs, ys, _, _ = state_space_data(T3, p3, 5, seed=DATA_SEED)
    pr3, ss3, sig_up = bsts_priors(Xs, ys, 5)
    e3 = boom_amd.Engine(C3, seed=SAMPLER_SEED, device=device)
    e3.ss_set_data(ys, Xs, None)
    e3.set_priors(pr3["b"], pr3["ominv"], pr3["pi"], pr3["df"], pr3["sigma_guess"],
                  sigma_upper_limit=sig_up)
    e3.ss_set_local_level(ss3["level_df"], ss3["level_sigma_guess"], ss3["level_sigma_upper_limit"],
                          ss3["initial_state_mean"], ss3["initial_state_variance"],
                          ss3["initial_level_sigma"])
    e3.set_state(np.zeros(p3, np.uint8))
    e3.ss_sweep(200)             # burn-in: the launch capacity follows the models down to 16
    dts = []
    for _ in range(5):           # (median: a capacity change inside a pass costs a relaunch)
        t0 = time.perf_counter()
        e3.ss_sweep(200)
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    k3 = float(e3.get_states()[0].sum(1).mean())
    # the callers' loop on this path (bindings/boom/DeviceStateSpacePosteriorSampler::draw):
    # one round per call, then chain 0's regression draw, level variance and state
    t0 = time.perf_counter()
    for _ in range(200):
        e3.ss_sweep(1)
        e3.get_state(0)
        e3.ss_get_state(0)
    loop3 = (time.perf_counter() - t0) / 200
    # ... and the same loop with the look-ahead the bindings switch on by default
    # (ba_ss_set_lookahead(64) + ba_ss_draw_next: batches of 64 rounds enqueued ahead, every
    # round's draw recorded on the device): median and quartiles over 20 batches of 64 calls
    e3.ss_set_lookahead(64)
    for _ in range(64):
        e3.ss_draw_next()
    per_batch = []
    for _ in range(20):
        t0 = time.perf_counter()
        for _ in range(64):
            e3.ss_draw_next()
            e3.get_state(0)
            e3.ss_get_state(0, suf=False)
        per_batch.append((time.perf_counter() - t0) / 64)
    q1, med, q3 = (float(v) for v in np.percentile(per_batch, [25, 50, 75]))
    e3.ss_set_lookahead(1)
    e3.set_kernel_timing(True)
    e3.ss_sweep(128)
    raw3 = e3.kernel_times()
    e3.set_kernel_timing(False)
    # the rounds of a call are ONE persistent launch per 64 of them (ss_round_kernel.hip):
    # device time per round = the launches' time over the rounds they ran
    kt = {k: round(ms / 128 * 1e3, 2) for k, (ms, _) in raw3.items()}       # us per round
    # ... and the separate launches of rounds 1-4 (ba_ss_set_tuning(4)) beside it, same engine
    e3.ss_set_tuning(kernel=4)
    e3.ss_sweep(20)
    t0 = time.perf_counter()
    e3.ss_sweep(200)
    dt_sep = time.perf_counter() - t0
    e3.set_kernel_timing(True)
    e3.ss_sweep(100)
    kt_sep = _per_launch(e3.kernel_times())
    e3.set_kernel_timing(False)
    e3.ss_set_tuning(kernel=5)
    # Kalman bytes per chain-sweep f (T + 2T); the regression half reads the
    # shared design matrix once per launch: T p f
    bytes3 = C3 * 3 * T3 * 8 + T3 * p3 * 8
    round_us = kt.get("ss_round_kernel", dt / 200 * 1e6)
    rec = {"sweeps_per_s": round(C3 * 200 / dt, 1), "us_per_round": round(dt / 200 * 1e6, 1),
           "mean_model_size": round(k3, 2), "kernel_us_per_round": kt,
           "separate_launches": {"what": "the same rounds as three launches each (regression sweep, state draw, "
                                         "X'e GEMM; ba_ss_set_tuning(4)): rounds 1-4's path, the fallback for series "
                                         "beyond 2048 steps and models beyond 48 variables",
                                 "us_per_round": round(dt_sep / 200 * 1e6, 1), "kernel_us_per_launch": kt_sep},
           "callers_loop_us_per_draw": round(med * 1e6, 1),
           "callers_loop_detail": {"lookahead": 64, "median_us": round(med * 1e6, 1),
                                   "iqr_us": [round(q1 * 1e6, 1), round(q3 * 1e6, 1)],
                                   "one_round_per_call_us": round(loop3 * 1e6, 1),
                                   "what": "draw_next(); get_state(0); ss_get_state(0) -- chain 0's "
                                           "regression draw, level variance and state path per draw"},
           "roofline": {"bound": "hbm", "kernel": "ss_round_kernel (every chain's rounds of a call in one persistent "
                                                   "launch: sweep, level variance + normals, state draw, X'e tiles)",
                        "algorithmic_bytes_per_round": bytes3,
                        "achieved": round(bytes3 / (round_us * 1e-6) / 1e9, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(bytes3 / (round_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        "round_frac": round(bytes3 / dt * 200 / 1e9 / HBM_PEAK_GBS, 4),
                        "state_draw_alone": {"kernel": "kalman_lm_kernel as its own launch (separate launches)",
                                             "us": kt_sep.get("kalman_simsmooth_kernel"),
                                             "frac": round(bytes3 / (kt_sep.get("kalman_simsmooth_kernel", 1e9) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                        "traffic": (lambda t: None if t is None else round(t / 64.0, 0))(_profile_traffic("c3", "ss_round_kernel")),
                        "traffic_source": "profiles/r*_c3_pmc_traffic.json (rocprofv3 --pmc passes of "
                                          "tools/ss_bench.py: launches of 64 rounds), per ROUND like `achieved`",
                        "note": "a round is bound by the two wavefronts' instruction streams (Philox for 2 T normals, "
                                "the sweep's dependent round trips), not by bandwidth: DESIGN 3.5"}}
    rec["roofline"] = _rebase(rec["roofline"], "c3", "ss_round_kernel", C3 * 200 / dt, "sweeps/s", 2)
    if cpu:
        opts3 = ssvs_options(sigma_upper_limit=sig_up)
        g3 = np.zeros(p3, np.uint8)
        rate, nsw = _cpu_rate(lambda c, n: O.ss_run(ys, Xs, None, pr3, opts3, ss3,
                                                    ("philox", SAMPLER_SEED, c), g3, n),
                              cores, cores)
        rec["cpu_baseline"] = {"value": round(rate, 2), "unit": "sweeps/s", "cores": cores, "kind": "port",
                               "sample": "%d chains x %d sweeps from the empty model on %d threads "
                                         "(oracle bo_ss_draw), same T=2000 p=100 data" % (cores, nsw, cores)}
    other["configs[2] bsts local level + regression T=2000 p=100, 1024 chains"] = rec
    e3.close()

    # ---- configs[3] per GPU: n=1e5 p=4096, 8192 chains / 8 GPUs = 1024 (the design matrix
    # is drawn on the device: 3.3 GB; X'X by the MFMA syrk; 32 signals) --------------------
    n4, p4, sig4, C4 = 100000, 4096, 32, 1024
    gen = torch.Generator(device="cuda")
    gen.manual_seed(DATA_SEED)
    X4 = torch.randn((p4, n4), dtype=torch.float64, device="cuda", generator=gen)   # column-major n x p
    X4[0].fill_(1.0)
    b4 = torch.zeros(p4, dtype=torch.float64, device="cuda")
    b4[:sig4] = torch.tensor([(1.0 + 0.1 * (i % 7)) * (-1.0) ** i for i in range(sig4)],
                             dtype=torch.float64, device="cuda")
    y4 = (b4[:sig4, None] * X4[:sig4]).sum(0) + torch.randn(n4, dtype=torch.float64, device="cuda", generator=gen)
    torch.cuda.synchronize()
    e4 = boom_amd.Engine(C4, seed=SAMPLER_SEED, device=device)
    e4.set_kernel_timing(True)
    t0 = time.perf_counter()
    e4.build_suf_from_xy_device(n4, p4, X4.data_ptr(), y4.data_ptr())
    e4.sync()
    build4 = time.perf_counter() - t0
    suf_ms = e4.kernel_times()["xtx_mfma_kernel+plane_sum_kernel+col_reduce_kernel"][0]
    e4.set_kernel_timing(False)
    del X4
    s4 = e4.get_suf()
    suf4 = dict(xtx=s4["xtx"], xty=s4["xty"], yty=s4["yty"], n=s4["n"], sumy=s4["ybar"] * s4["n"],
                xsum=s4["xbar"] * s4["n"])
    pr4 = spike_slab_prior(suf4, sig4)
    e4.set_priors(pr4["b"], pr4["ominv"], pr4["pi"], pr4["df"], pr4["sigma_guess"])
    g4 = np.zeros(p4, np.uint8)
    g4[0] = 1
    e4.set_state(g4)
    e4.sweep(60)
    e4.reset_summaries()
    t0 = time.perf_counter()
    e4.sweep(40)
    dt = time.perf_counter() - t0
    sm4 = e4.get_summaries()
    k4 = sm4["k_sum"] / sm4["sweeps"]
    gam4, beta4, sig4v = e4.get_states()
    e4.set_kernel_timing(True)
    e4.sweep(40)
    kt = e4.kernel_times()
    e4.set_kernel_timing(False)
    ms40 = sum(ms for ms, _ in kt.values())      # (the LDS kernel, and the large-model one if any chain needed it)
    bytes4 = (p4 * 8.0 * (2 * k4 + 4) + 8.0 * (3 * k4 + 4) + p4 / 8.0) * C4 * 40
    rec = {"sweeps_per_s": round(C4 * 40 / dt, 1), "ms_per_round": round(dt / 40 * 1e3, 3),
           "suf_build_ms": round(build4 * 1e3, 1),
           "mean_model_size": round(float(k4), 2),
           "signal_inclusion_min": round(float(gam4[:, :sig4].mean(0).min()), 4),
           "kernel_ms_per_40_sweep_launch": {k: round(ms, 3) for k, (ms, _) in kt.items()},
           "roofline": {"bound": "hbm", "kernel": "ssvs_sweep_kernel",
                        "algorithmic_bytes_per_launch": round(bytes4, 0),
                        "achieved": round(bytes4 / (ms40 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(bytes4 / (ms40 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": _profile_traffic("c4", "ssvs_sweep_kernel"),
                        "traffic_source": "profiles/r*_c4_pmc_traffic.json (rocprofv3 --pmc passes of "
                                          "`bench.py --config 3`, tools/regen_profiles.sh), per 40-sweep launch"},
           "suf_roofline": {"bound": "mfma", "kernel": "xtx_mfma_kernel", "ms": round(suf_ms, 3),
                            "achieved": round(n4 * float(p4) * p4 / (suf_ms * 1e-3) / 1e12, 2),
                            "peak": F64_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                            "frac": round(n4 * float(p4) * p4 / (suf_ms * 1e-3) / 1e12 / F64_MATRIX_PEAK_TF, 4),
                            "note": "flops = n p^2 (SURVEY 8d: syrk half); the whole build incl. X'y and sums"}}
    rec["roofline"] = _rebase(rec["roofline"], "c4", "ssvs_sweep_kernel", C4 * 40 / dt, "sweeps/s", 2)
    if cpu:
        def run4(nchains, nsw, nthreads):
            t0 = time.perf_counter()
            O.run_chains(suf4, pr4, ssvs_options(), SAMPLER_SEED, nchains, nsw, nthreads,
                         gam4[0], beta4[0], float(sig4v[0]))
            return nchains * nsw / (time.perf_counter() - t0)
        cal = run4(cores, 2, cores)
        nsw = int(max(2, min(400, 6.0 * cal / cores)))
        # (the sample is bounded by the default run's budget, so it is taken
        # twice and the line carries the spread of the two)
        r4a, r4b = run4(cores, nsw, cores), run4(cores, nsw, cores)
        rate = 0.5 * (r4a + r4b)
        rec["cpu_baseline"] = {"value": round(rate, 2), "unit": "sweeps/s", "cores": cores, "kind": "port",
                               "spread_rel": round(abs(r4a - r4b) / rate, 3),
                               "sample": "2 x (%d chains x %d sweeps) on %d pthreads, warm-started at a GPU chain's "
                                         "state (kbar~%.1f), same n=1e5 p=4096 statistics; value = mean of the two "
                                         "runs, spread_rel = their difference / mean" % (cores, nsw, cores, k4)}
    other["configs[3] per GPU: spike-and-slab n=1e5 p=4096, 1024 chains"] = rec
    e4.close()
    del s4, suf4, pr4
    torch.cuda.empty_cache()

    # ---- configs[4] per GPU: logit spike-and-slab n=5e4 p=1024, 4096 chains / 8 GPUs = 512
    # (the reference's auxiliary-mixture imputer; it has no Polya-Gamma sampler) ----------
    n5, p5, C5 = 50000, 1024, 512
    Xl, yl, ntl, _ = logit_data(n5, p5, 8, seed=DATA_SEED)
    slab5, pi5 = probit_slab(Xl, ntl, 8)
    e5 = boom_amd.Engine(C5, seed=SAMPLER_SEED, device=device)
    e5.logit_set_data(Xl, yl, ntl, 5)
    e5.sss_set_slab(slab5["mu"], slab5["prec"], scales_with_sigsq=False)
    e5.set_spike(pi5)
    g5 = np.zeros(p5, np.uint8)
    g5[0] = 1
    e5.set_state(g5)
    e5.logit_sweep(15)
    t0 = time.perf_counter()
    e5.logit_sweep(30)
    dt = time.perf_counter() - t0
    gam5 = e5.get_states()[0]
    k5 = float(gam5.sum(1).mean())
    e5.set_kernel_timing(True)
    e5.logit_sweep(10)
    kt = {k: round(ms / 10, 3) for k, (ms, _) in e5.kernel_times().items()}    # ms per round
    e5.set_kernel_timing(False)
    cols = "xtwx_cols_kernel<true>+xtwx_cols_reduce_kernel"
    flops5 = 2.0 * n5 * p5 * k5 * C5       # the request GEMM: R = sum of model sizes rows of p, n deep
    rec = {"sweeps_per_s": round(C5 * 30 / dt, 1), "ms_per_round": round(dt / 30 * 1e3, 2),
           "mean_model_size": round(k5, 2),
           "signal_inclusion_min": round(float(gam5[:, :8].mean(0).min()), 4),
           "kernel_ms_per_round": kt,
           "roofline": {"bound": "mfma", "kernel": "xtwx_cols_kernel<true>",
                        "flops_per_round": flops5, "achieved": round(flops5 / (kt[cols] * 1e-3) / 1e12, 2),
                        "peak": F64_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(flops5 / (kt[cols] * 1e-3) / 1e12 / F64_MATRIX_PEAK_TF, 4),
                        "traffic": _profile_traffic("logit", "xtwx_cols_kernel<true"),
                        "traffic_source": "profiles/r*_logit_pmc_traffic.json, per launch of the request GEMM "
                                          "(several launches a round)",
                        "note": "v_mfma_f64_16x16x4_f64; R taken as chains x mean model size (the vectors "
                                "requested mid-sweep add a few percent)"}}
    if cpu:
        # ALL 50 000 observations, one sweep of one chain per thread (a sweep of the reference
        # algorithm is dominated by the n p^2 rebuild of X'WX: about a minute per chain)
        th = min(cores, 8)
        rate, nsw = _cpu_rate(lambda c, n: O.logit_run(Xl, yl, ntl, slab5, pi5,
                                                       ("philox", SAMPLER_SEED, c), g5, np.zeros(p5), n),
                              th, th, target_s=1.0, first=1)
        rec["cpu_baseline"] = {"value": round(rate, 3), "unit": "sweeps/s", "cores": th,
                               "kind": "port", "sample_seconds": round(th * nsw / rate, 1),
                               "sample": "%d chains x %d sweep(s) on %d threads on all %d observations "
                                         "(oracle bo_logit_draw from the one-variable model); the threads run "
                                         "the same work on different streams, so the elapsed time is the "
                                         "slowest thread's: a lower bound of the rate, within a few percent"
                                         % (th, nsw, th, n5)}
        nsub = 5000
        Xsub, ysub, ntsub = np.ascontiguousarray(Xl[:nsub]), yl[:nsub], ntl[:nsub]
        slabs, pis = probit_slab(Xsub, ntsub, 8)
    other["configs[4] per GPU: logit spike-and-slab n=5e4 p=1024, 512 chains"] = rec
    # ... and with the imputer BASELINE words the configuration with: Polya-Gamma augmentation
    # (ba_logit_set_imputer(1); BOOM has no such sampler -- parity for it is distributional,
    # tests/test_polya_gamma.py), same data, same chains, continuing from the state above
    e5.logit_set_imputer(1)
    e5.logit_sweep(10)
    t0 = time.perf_counter()
    e5.logit_sweep(30)
    dtp = time.perf_counter() - t0
    gamp = e5.get_states()[0]
    e5.set_kernel_timing(True)
    e5.logit_sweep(10)
    ktp = {k: round(ms / 10, 3) for k, (ms, _) in e5.kernel_times().items()}
    e5.set_kernel_timing(False)
    kp = float(gamp.sum(1).mean())
    flopsp = 2.0 * n5 * p5 * kp * C5
    other["configs[4] per GPU with the Polya-Gamma imputer"] = {
        "sweeps_per_s": round(C5 * 30 / dtp, 1), "ms_per_round": round(dtp / 30 * 1e3, 2),
        "mean_model_size": round(kp, 2),
        "signal_inclusion_min": round(float(gamp[:, :8].mean(0).min()), 4),
        "kernel_ms_per_round": ktp,
        "roofline": {"bound": "mfma", "kernel": "xtwx_cols_kernel<true>", "flops_per_round": flopsp,
                     "achieved": round(flopsp / (ktp[cols] * 1e-3) / 1e12, 2), "peak": F64_MATRIX_PEAK_TF,
                     "unit": "TFLOP/s", "frac": round(flopsp / (ktp[cols] * 1e-3) / 1e12 / F64_MATRIX_PEAK_TF, 4),
                     "traffic": _profile_traffic("pg", "xtwx_cols_kernel<true"),
                     "traffic_source": "profiles/r*_pg_pmc_traffic.json, per launch of the request GEMM"},
        "cpu_baseline": None}
    if cpu:
        rate, nsw = _cpu_rate(lambda c, n: O.logit_run(Xsub, ysub, ntsub, slabs, pis,
                                                       ("philox", SAMPLER_SEED, c), g5, np.zeros(p5), n, imputer=1),
                              th, th, target_s=10.0, first=1)
        other["configs[4] per GPU with the Polya-Gamma imputer"]["cpu_baseline"] = {
            "value": round(rate * nsub / n5, 3), "unit": "sweeps/s", "cores": th, "kind": "port",
            "sample": "%d chains x %d sweeps on %d threads on the FIRST %d of the %d observations with the "
                      "oracle's Polya-Gamma imputer (bo_logit_set_imputer(1)), rate scaled by %d/%d as for the "
                      "auxiliary-mixture line" % (th, nsw, th, nsub, n5, nsub, n5)}
    e5.close()

    # ---- SURVEY 8(d)'s dense-posterior variant of configs[1]: the same n=1e4, p=512, 1024
    # chains with 64 true signals (the models sit at the LDS kernel's 64-variable limit)
    from cases import regression_data
    nd, pd_, sigd, Cd = 10000, 512, 64, 1024
    Xd, yd, _ = regression_data(nd, pd_, sigd, seed=DATA_SEED)
    ed = boom_amd.Engine(Cd, seed=SAMPLER_SEED, device=device)
    ed.build_suf_from_xy(Xd, yd)
    sd = ed.get_suf()
    sufd = dict(xtx=sd["xtx"], xty=sd["xty"], yty=sd["yty"], n=sd["n"], sumy=sd["ybar"] * sd["n"],
                xsum=sd["xbar"] * sd["n"])
    prd = spike_slab_prior(sufd, sigd)
    ed.set_priors(prd["b"], prd["ominv"], prd["pi"], prd["df"], prd["sigma_guess"])
    gd = np.zeros(pd_, np.uint8)
    gd[0] = 1
    ed.set_state(gd)
    # (the headline's step: launches of SWEEPS_PER_STEP sweeps -- a launch lasts as long as its
    # slowest chain, and over 200 sweeps the slowest of 1024 chains makes 1.4x the mean's
    # accepted flips, over 1000 sweeps 1.15x)
    NSD = SWEEPS_PER_STEP
    ed.sweep(200)
    ed.reset_summaries()
    t0 = time.perf_counter()
    ed.sweep(NSD)
    dtd = time.perf_counter() - t0
    smd = ed.get_summaries()
    kd = smd["k_sum"] / smd["sweeps"]
    gamd, betad, sigdv = ed.get_states()
    ed.set_kernel_timing(True)
    ed.sweep(NSD)
    ktd = ed.kernel_times()
    ed.set_kernel_timing(False)
    msd = sum(ms for ms, _ in ktd.values())
    bytesd = (pd_ * 8.0 * (2 * kd + 4) + 8.0 * (3 * kd + 4) + pd_ / 8.0) * Cd * NSD
    recd = {"sweeps_per_s": round(Cd * NSD / dtd, 1), "us_per_sweep_round": round(dtd / NSD * 1e6, 1),
            "mean_model_size": round(float(kd), 2),
            "signal_inclusion_min": round(float(gamd[:, :sigd].mean(0).min()), 4),
            "kernel_ms_per_%d_sweep_launch" % NSD: {k: round(ms, 3) for k, (ms, _) in ktd.items()},
            "roofline": {"bound": "hbm", "kernel": max(ktd, key=lambda k: ktd[k][0]),
                         "algorithmic_bytes_per_launch": round(bytesd, 0),
                         "achieved": round(bytesd / (msd * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(bytesd / (msd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": _profile_traffic("dense64", "ssvs_big_kernel"),
                         "traffic_source": "profiles/r*_dense64_pmc_traffic.json, per %d-sweep launch of ssvs_big_kernel "
                                           "(NSWEEP=%d tools/dense_variant.py 64)" % (NSD, NSD)}}
    recd["roofline"]["note"] = ("models of 64-66 variables: factors in HBM, streamed per fill; the kernel waits (wait_any 0.79 "
                                "in round 5's counters) on those streams and on its scratch")
    recd["roofline"] = _rebase(recd["roofline"], "dense64", "ssvs_big_kernel", Cd * NSD / dtd, "sweeps/s", 2)
    if cpu:
        def rund(nchains, nsw, nthreads):
            t0 = time.perf_counter()
            O.run_chains(sufd, prd, ssvs_options(), SAMPLER_SEED, nchains, nsw, nthreads,
                         gamd[0], betad[0], float(sigdv[0]))
            return nchains * nsw / (time.perf_counter() - t0)
        cal = rund(cores, 10, cores)
        nsw = int(max(10, min(2000, 6.0 * cal / cores)))
        rate = rund(cores, nsw, cores)
        recd["cpu_baseline"] = {"value": round(rate, 2), "unit": "sweeps/s", "cores": cores, "kind": "port",
                                "sample": "%d chains x %d sweeps on %d pthreads, warm-started at a GPU chain's "
                                          "state (kbar~%.1f), same statistics" % (cores, nsw, cores, kd)}
    other["dense_variant_64_signals (configs[1] data shape, 64 true signals)"] = recd
    ed.close()
    other.update(family_configs(boom_amd, device, O, cores, cpu))
    return other


def family_configs(boom_amd, device, O, cores, cpu):
    """SURVEY 8(f)'s rows at the BASELINE shapes: the structural state
    models (f2: the shapes the specialised kernel covers and two lists it does not), the
    adaptive sampler (f1) on configs[1]'s data, the probit and Poisson samplers (f3) at
    configs[4]'s per-GPU shape.  Each: sweeps/s, device time of its kernels per round, the
    roofline of its dominant kernel (algorithmic bytes per DESIGN 3.x over that kernel's
    time), and the oracle timed on this box's cores on a bounded sample of the same workload."""
    from cases import (bsts_priors, general_data, general_spec, poisson_data, probit_data, probit_slab,
                       regression_data, spike_slab_prior, structural_data, structural_spec)
    from oracle_lib import ssvs_options
    out = {}
    T, p, nsig, C = 2000, 100, 5, 1024

    def state_space(name, blocks_desc, template, ar_coef, tag, nrounds):
        seas = [(d[1], d[2]) for d in blocks_desc if d[0] == "seasonal"]
        X, y, _, _ = general_data(T, p, nsig, seas[:2], seed=DATA_SEED, ar_coef=ar_coef)
        prior, _, sig_up = bsts_priors(X, y, 5)
        blocks = general_spec(y, blocks_desc)
        m = sum(bl["dim"] for bl in blocks)
        nvar = sum(len(bl["df"]) for bl in blocks)
        eng = boom_amd.Engine(C, seed=SAMPLER_SEED, device=device)
        eng.ss_set_data(y, X, None)
        eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"],
                       sigma_upper_limit=sig_up)
        eng.ss_set_state_models(blocks)
        eng.set_state(np.zeros(p, np.uint8))
        eng.ss_sweep(max(6, nrounds // 2))
        t0 = time.perf_counter()
        eng.ss_sweep(nrounds)
        dt = (time.perf_counter() - t0) / nrounds
        eng.set_kernel_timing(True)
        eng.ss_sweep(nrounds)
        kt = {k: round(ms / nrounds * 1e3, 1) for k, (ms, _) in eng.kernel_times().items()}   # us per round
        eng.set_kernel_timing(False)
        kbar = float(eng.get_states()[0].sum(1).mean())
        eng.close()
        # per chain-sweep T f (2 m + 3 + nper + 6): gains and state (m each), three
        # rows of smoothed disturbances, the step's normals, y*, F, residuals
        bytes_ = float(C) * T * 8 * (2 * m + 3 + (nvar + 1) + 6)
        kern = "ssm_simsmooth_kernel"
        us = kt.get(kern, max(kt.values()))
        rec = {"state_dimension": m, "kernel": "ssm_simsmooth_kernel<...> (shape-specialised)" if template
               else "ssg_simsmooth_kernel (any list of state models)",
               "sweeps_per_s": round(C / dt, 1), "ms_per_round": round(dt * 1e3, 3), "mean_model_size": round(kbar, 2),
               "kernel_us_per_round": kt,
               "roofline": {"bound": "hbm", "kernel": kern + " timing class (the state draw)",
                            "algorithmic_bytes_per_round": bytes_,
                            "achieved": round(bytes_ / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(bytes_ / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                            "traffic": _profile_traffic(tag, "ssm_simsmooth_kernel" if template else "ssg_simsmooth_kernel"),
                            "traffic_source": "profiles/r*_%s_pmc_traffic.json, per launch of the state draw" % tag,
                            "note": "serial in time: bound by the dependent-instruction chain of a step, not "
                                    "by bandwidth (DESIGN 3.6)"}}
        rec["roofline"] = _rebase(rec["roofline"], tag, "ssm_simsmooth_kernel" if template else "ssg_simsmooth_kernel",
                                  C / dt, "sweeps/s", 2)
        if cpu:
            opts = ssvs_options(sigma_upper_limit=sig_up)
            g0 = np.zeros(p, np.uint8)
            rate, nsw = _cpu_rate(lambda c, n: O.ssg_run(y, X, None, prior, opts, blocks, ("philox", SAMPLER_SEED, c),
                                                         g0, n, state_every=10 ** 6),
                                  cores, cores, target_s=5.0, first=2)
            rec["cpu_baseline"] = {"value": round(rate, 2), "unit": "sweeps/s", "cores": cores, "kind": "port",
                                   "sample": "%d chains x %d sweeps from the empty model on %d threads (oracle "
                                             "bo_ssm_draw), same T=2000 p=100 data and state models"
                                             % (cores, nsw, cores)}
        out[name] = rec

    state_space("f2 structural: trend + 12 seasons (m=13), T=2000 p=100, 1024 chains",
                [("trend",), ("seasonal", 12, 1)], True, None, "structural", 30)
    state_space("f2 structural: trend + 12 seasons + AR(2) (m=15), T=2000 p=100, 1024 chains",
                [("trend",), ("seasonal", 12, 1), ("ar", 2)], True, [1.2, -0.4], "structural_ar", 30)
    state_space("f2 general list: trend + weekly + 4 x 7 cycle (m=11), T=2000 p=100, 1024 chains",
                [("trend",), ("seasonal", 7, 1), ("seasonal", 4, 7)], False, None, "general", 16)
    state_space("f2 general list: bsts daily model, trend + weekly + 52 x 7 cycle (m=59), T=2000 p=100, 1024 chains",
                [("trend",), ("seasonal", 7, 1), ("seasonal", 52, 7)], False, None, "general", 8)

    # ---- f1: AdaptiveSpikeSlabRegressionSampler on configs[1]'s data (what lm.spike runs for p > 100)
    n2, p2, sig2, C2 = 10000, 512, 16, 1024
    X2, y2, _ = regression_data(n2, p2, sig2, seed=DATA_SEED)
    ea = boom_amd.Engine(C2, seed=SAMPLER_SEED, device=device)
    ea.build_suf_from_xy(X2, y2)
    s2 = ea.get_suf()
    suf2 = dict(xtx=s2["xtx"], xty=s2["xty"], yty=s2["yty"], n=s2["n"], sumy=s2["ybar"] * s2["n"],
                xsum=s2["xbar"] * s2["n"])
    pr2 = spike_slab_prior(suf2, sig2)
    ea.set_priors(pr2["b"], pr2["ominv"], pr2["pi"], pr2["df"], pr2["sigma_guess"])
    ga = np.zeros(p2, np.uint8)
    ga[0] = 1
    ea.set_state(ga)
    ea.adaptive_sweep(300)
    ea.reset_summaries()
    dta = float("inf")
    for _ in range(3):   # (22 ms a call: the best of three -- one host hiccup tripled a single shot)
        t0 = time.perf_counter()
        ea.adaptive_sweep(500)
        dta = min(dta, time.perf_counter() - t0)
    sma = ea.get_summaries()
    ka = sma["k_sum"] / sma["sweeps"]
    ea.set_kernel_timing(True)
    ea.adaptive_sweep(500)
    kta = ea.kernel_times()
    ea.set_kernel_timing(False)
    gama, betaa, siga = ea.get_states()
    ea.close()
    msa = sum(ms for ms, _ in kta.values())
    # the reference's iteration: max_flips (100) birth / death proposals, each a log_model_prob
    # of a model one variable away: 100 x f (2 k + 4) gathered + the draw's 3 k + 4
    bytesa = (100 * 8.0 * (2 * ka + 4) + 8.0 * (3 * ka + 4)) * C2 * 500
    reca = {"sweeps_per_s": round(C2 * 500 / dta, 1), "us_per_sweep_round": round(dta / 500 * 1e6, 1),
            "mean_model_size": round(float(ka), 2), "accepted_moves_per_sweep": round(sma["accepts"] / sma["sweeps"], 3),
            "kernel_ms_per_500_sweep_launch": {k: round(ms, 3) for k, (ms, _) in kta.items()},
            "roofline": {"bound": "hbm", "kernel": "ssvs_adaptive_kernel", "algorithmic_bytes_per_launch": round(bytesa, 0),
                         "achieved": round(bytesa / (msa * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(bytesa / (msa * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "note": "100 proposals a sweep instead of p = 512: a fifth of the headline's bytes per sweep"}}
    reca["roofline"] = _rebase(reca["roofline"], "adaptive", "ssvs_adaptive_kernel", C2 * 500 / dta, "sweeps/s", 2)
    if cpu:
        rate, nsw = _cpu_rate(lambda c, n: O.adaptive_run(suf2, pr2, ssvs_options(), ("philox", SAMPLER_SEED, c), gama[c % C2], n),
                              cores, cores, target_s=5.0, first=20)
        reca["cpu_baseline"] = {"value": round(rate, 2), "unit": "sweeps/s", "cores": cores, "kind": "port",
                                "sample": "%d chains x %d sweeps on %d threads, started at GPU chains' models "
                                          "(kbar~%.1f), same statistics (oracle bo_adaptive_draw)" % (cores, nsw, cores, ka)}
    out["f1 adaptive sampler (AdaptiveSpikeSlabRegressionSampler), configs[1]'s data: n=1e4 p=512, 1024 chains"] = reca

    # ---- f3: probit and Poisson at configs[4]'s per-GPU shape (n = 5e4, p = 1024, 512 chains)
    n5, p5, C5 = 50000, 1024, 512

    def glm(name, kind):
        if kind == "probit":
            X, y, nt, _ = probit_data(n5, p5, 8, seed=DATA_SEED)
            slab, pi = probit_slab(X, nt, 8)
        else:
            from test_oracle_golden import _golden_mix, load
            mix = _golden_mix(load("poisson_small_counts"))
            X, y, ex, _ = poisson_data(n5, p5, 8, seed=DATA_SEED)
            y = np.minimum(y, 11.0)      # (the mixture table that travels with the goldens covers counts 1 .. 11)
            slab, pi = probit_slab(X, np.ones(n5), 8)
        e = boom_amd.Engine(C5, seed=SAMPLER_SEED, device=device)
        if kind == "probit":
            e.probit_set_data(X, y, nt, 5)
            sweep = e.probit_sweep
        else:
            e.poisson_set_data(X, y, ex, mix)
            sweep = e.poisson_sweep
        e.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
        e.set_spike(pi)
        g0 = np.zeros(p5, np.uint8)
        g0[0] = 1
        e.set_state(g0)
        sweep(10)
        t0 = time.perf_counter()
        sweep(20)
        dt = (time.perf_counter() - t0) / 20
        e.set_kernel_timing(True)
        sweep(10)
        kt = {k: round(ms / 10, 3) for k, (ms, _) in e.kernel_times().items()}
        e.set_kernel_timing(False)
        gam = e.get_states()[0]
        kbar = float(gam.sum(1).mean())
        e.close()
        imp = "probit_impute_kernel" if kind == "probit" else "poisson_impute_kernel"
        # the imputation reads, per chain and observation, the included variables' entries of
        # the row, y / trials / exposure, and writes the latent value(s)
        bytes_ = float(C5) * n5 * 8.0 * (kbar + (3 if kind == "probit" else 5))
        rec = {"sweeps_per_s": round(C5 / dt, 1), "ms_per_round": round(dt * 1e3, 2), "mean_model_size": round(kbar, 2),
               "signal_inclusion_min": round(float(gam[:, :8].mean(0).min()), 4), "kernel_ms_per_round": kt,
               "roofline": {"bound": "hbm", "kernel": imp, "algorithmic_bytes_per_round": bytes_,
                            "achieved": round(bytes_ / (kt[imp] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(bytes_ / (kt[imp] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "traffic": _profile_traffic(kind, imp),
                            "traffic_source": "profiles/r*_%s_pmc_traffic.json, per launch of the imputation" % kind,
                            "note": "rejection / adaptive-rejection loops per observation: bound by their latency "
                                    "and divergence, not by bandwidth (DESIGN 3.8)"}}
        rec["roofline"] = _rebase(rec["roofline"], kind, imp, C5 / dt, "sweeps/s", 4)
        if cpu:
            th = min(cores, 8)
            if kind == "probit":
                fn = lambda c, n: O.probit_run(X, y, nt, slab, pi, ("philox", SAMPLER_SEED, c), g0, np.zeros(p5), n)
            else:
                fn = lambda c, n: O.poisson_run(X, y, ex, slab, pi, mix, ("philox", SAMPLER_SEED, c), g0, np.zeros(p5), n)
            rate, nsw = _cpu_rate(fn, th, th, target_s=8.0, first=1)
            rec["cpu_baseline"] = {"value": round(rate, 3), "unit": "sweeps/s", "cores": th, "kind": "port",
                                   "sample": "%d chains x %d sweeps on %d threads on all %d observations" % (th, nsw, th, n5)}
        out[name] = rec

    glm("f3 probit spike-and-slab at configs[4]'s per-GPU shape: n=5e4 p=1024, 512 chains", "probit")
    glm("f3 Poisson spike-and-slab at configs[4]'s per-GPU shape: n=5e4 p=1024, 512 chains (counts capped at 11: "
        "the goldens' mixture table)", "poisson")
    return out


def _timed_steps(torch, dist, eng, world, local_rank, steps, warmup, step):
    """W untimed + K timed steps between barriers; returns (wall seconds, ms per step by
    HIP events on the engine's stream)"""
    est = torch.cuda.ExternalStream(eng.stream(), device=torch.device("cuda", local_rank))
    for _ in range(warmup):
        step()
    eng.sync()
    eng.reset_summaries()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(est)
    for _ in range(steps):
        step()
        eng.stream()       # (steps kept apart, as in the headline)
    ev1.record(est)
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1) / max(1, steps)


def run_config3(args, boom_amd, torch, dist, rank, local_rank, world):
    """BASELINE configs[3]: spike-and-slab n=1e5 p=4096, 1024 chains per GPU.  SURVEY 8(e):
    the design matrix is ROW-SHARDED -- every rank draws its own rows on its device (3.3 GB
    over the job, never through the host) --, each rank runs the MFMA syrk on its rows, ONE
    all-reduce of the (p^2 + 2p + 2)-double block (134 MB) over RCCL, every rank installs
    the bitwise-identical total; the chains never communicate; one all-gather of the
    posterior summaries at the end.  A step = one launch of 40 sweeps of every chain."""
    from boom_amd import dist as bd
    from cases import spike_slab_prior
    n, p, C = args.n_obs or 100000, args.p or 4096, args.chains or 1024
    nsig, SW, BURN = 32, 40, 60
    lo, hi = bd.row_shard(n, rank, world)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(DATA_SEED + 104729 * rank)      # (a rank's rows are its own)
    Xs = torch.randn((p, hi - lo), dtype=torch.float64, device="cuda", generator=gen)   # column-major rows x p
    Xs[0].fill_(1.0)
    b = torch.zeros(p, dtype=torch.float64, device="cuda")
    b[:nsig] = torch.tensor([(1.0 + 0.1 * (i % 7)) * (-1.0) ** i for i in range(nsig)],
                            dtype=torch.float64, device="cuda")
    ys = (b[:nsig, None] * Xs[:nsig]).sum(0) + torch.randn(hi - lo, dtype=torch.float64, device="cuda", generator=gen)
    torch.cuda.synchronize()
    eng = boom_amd.Engine(C, seed=SAMPLER_SEED, device=local_rank, chain_offset=rank * C)
    t0 = time.perf_counter()
    if world > 1:
        block = bd.build_suf_row_sharded(eng, Xs, ys, n, world)
    else:
        eng.build_suf_from_xy_device(n, p, Xs.data_ptr(), ys.data_ptr())
        block = None
    eng.sync()
    suf_build_s = time.perf_counter() - t0
    del Xs, ys
    s = eng.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"], sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    prior = spike_slab_prior(suf, nsig)
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"], prior["sigma_guess"])
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.sweep(BURN)
    elapsed, kernel_ms = _timed_steps(torch, dist, eng, world, local_rank, args.steps, args.warmup,
                                      lambda: eng.sweep(SW, sync=False))
    blk = torch.empty(bd.summary_block_size(p), dtype=torch.float64, device="cuda")
    eng.summaries_device(blk.data_ptr())
    allb = bd.gather_blocks(blk, world)
    coll = _collectives(bd, dist, world)
    elapsed = bd.max_over_ranks(elapsed, world, "cuda")
    if args.dump_blocks:
        dig = torch.tensor([float(rank * C), float(s["xtx"].sum()), float(np.abs(s["xtx"]).sum()), float(s["xty"].sum()),
                            float(s["yty"]), float(s["ybar"]), float(s["xbar"].sum()), float(s["n"])],
                           dtype=torch.float64, device="cuda")
        digs = bd.gather_blocks(dig, world)
        if rank == 0:
            np.savez(args.dump_blocks, blocks=allb, digests=digs,
                     suf_block=(block.cpu().numpy() if block is not None else np.zeros(0)))
    if rank != 0:
        return
    sc = allb[:, 3 * p:]
    total = float(sc[:, 0].sum())
    kbar = float(sc[:, 3].sum() / total)
    incl = allb[:, :p].sum(axis=0) / total
    bytes_per_sweep = p * 8.0 * (2 * kbar + 4) + 8.0 * (3 * kbar + 4) + p / 8.0
    achieved = bytes_per_sweep * C * SW / (kernel_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "Gibbs sweeps/sec (all chains), n=%g p=%d spike-slab" % (n, p),
        "value": round(total / elapsed, 1), "unit": "sweeps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BregVsSampler spike-and-slab n=%g p=%d, %d chains per GPU, fp64 (BASELINE "
                               "configs[3]: 8192 chains over 8 GPUs)" % (n, p, C),
                   "chains_per_gpu": C, "sweeps_per_step": SW, "true_signals": nsig,
                   "mean_model_size": round(kbar, 2), "burn_in": BURN,
                   "parallelism": "chains sharded, %d GPU(s)" % world,
                   "suf_build": ("rows sharded (each rank draws its rows on its device), local MFMA syrk, ONE "
                                 "all-reduce of %d bytes" % (8 * bd.suf_block_size(p))) if world > 1
                                else "single device MFMA syrk"},
        "suf_build_ms": round(suf_build_s * 1e3, 1),
        "collectives": coll,
        "decisions": {"min_margin": float(sc[:, 6].min()), "accepted_flips": float(sc[:, 4].sum()),
                      "proposed_flips": float(sc[:, 5].sum())},
        "signal_inclusion_min": round(float(incl[:nsig].min()), 4),
        "roofline": _rebase({"kernel": "ssvs_sweep_kernel", "kernel_ms": round(kernel_ms, 4),
                             "algorithmic_bytes_per_sweep": round(bytes_per_sweep, 1), "achieved": round(achieved, 2),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                             "traffic": _profile_traffic("c4", "ssvs_sweep_kernel")},
                            "c4", "ssvs_sweep_kernel", total / elapsed / world, "sweeps/s (one GPU)", 2),
        "cpu_baseline": None}))


def run_config4(args, boom_amd, torch, dist, rank, local_rank, world):
    """BASELINE configs[4]: logit spike-and-slab n=5e4 p=1024, 512 chains per GPU (4096 over
    8 GPUs).  The data are replicated (every rank holds X: 410 MB), the chains sharded, no
    collective on the sampling path, one all-gather of the summaries at the end.  A step =
    5 sweep rounds (imputation, X'z, the requested vectors of X'WX, the sweep)."""
    from boom_amd import dist as bd
    from cases import logit_data, probit_slab
    n, p, C = args.n_obs or 50000, args.p or 1024, args.chains or 512
    R, BURN = 5, 15
    X, y, nt, _ = logit_data(n, p, 8, seed=DATA_SEED)
    slab, pi = probit_slab(X, nt, 8)
    eng = boom_amd.Engine(C, seed=SAMPLER_SEED, device=local_rank, chain_offset=rank * C)
    eng.logit_set_data(X, y, nt, 5)
    if args.imputer == "pg":
        eng.logit_set_imputer(1)
    eng.sss_set_slab(slab["mu"], slab["prec"], scales_with_sigsq=False)
    eng.set_spike(pi)
    g0 = np.zeros(p, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.logit_sweep(BURN)
    elapsed, step_ms = _timed_steps(torch, dist, eng, world, local_rank, args.steps, args.warmup,
                                    lambda: eng.logit_sweep(R, sync=False))
    blk = torch.empty(bd.summary_block_size(p), dtype=torch.float64, device="cuda")
    eng.summaries_device(blk.data_ptr())
    allb = bd.gather_blocks(blk, world)
    coll = _collectives(bd, dist, world)
    elapsed = bd.max_over_ranks(elapsed, world, "cuda")
    gam = eng.get_states()[0]
    if args.dump_blocks and rank == 0:
        np.savez(args.dump_blocks, blocks=allb)
    if rank != 0:
        return
    total = float(world * C * R * args.steps)
    counted = float(allb[:, 3 * p].sum())
    incl = allb[:, :p].sum(axis=0) / max(counted, 1.0)
    print(json.dumps({
        "metric": "Gibbs sweeps/sec (all chains), logit spike-slab n=%g p=%d" % (n, p),
        "value": round(total / elapsed, 1), "unit": "sweeps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BinomialLogitSpikeSlabSampler n=%g p=%d, %d chains per GPU, fp64 (BASELINE "
                               "configs[4]: 4096 chains over 8 GPUs), imputer: %s"
                               % (n, p, C, "Polya-Gamma" if args.imputer == "pg" else
                                  "auxiliary mixture (the reference's)"),
                   "chains_per_gpu": C, "rounds_per_step": R, "burn_in": BURN,
                   "mean_model_size": round(float(gam.sum(1).mean()), 2),
                   "parallelism": "chains sharded, data replicated, %d GPU(s)" % world},
        "sweeps_in_the_summaries": counted,
        "collectives": coll,
        "signal_inclusion_min": round(float(incl[:8].min()), 4),
        "ms_per_round": round(step_ms / R, 3),
        "cpu_baseline": None}))


def _collectives(bd, dist, world):
    """what the job's collectives were and took on rank 0 (ONE all-reduce of the
    sufficient-statistics block where the rows are sharded, ONE all-gather of the summary
    blocks at the end): backend, ranks under RCCL, wall ms (device-synchronised on both sides).
    Call right after the summary gather."""
    if world == 1:
        return {"backend": None, "rccl_ranks": 0}
    be = dist.get_backend()
    out = {"backend": be, "rccl_ranks": world if be == "nccl" else 0}
    for k in ("all_reduce_ms", "all_gather_ms"):
        if k in bd.timings:
            out[k] = round(bd.timings[k], 3)
            out[k.replace("_ms", "_bytes")] = int(bd.timings[k.replace("_ms", "_bytes")])
    return out


def _profile_traffic(tag, kernel_prefix):
    """HBM bytes per launch of a kernel from the newest committed PMC passes
    (profiles/r*_<tag>_pmc_traffic.json, tools/regen_profiles.sh), or None"""
    try:
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_pmc_traffic.json" % tag)))
        with open(cand[-1]) as fh:
            tj = json.load(fh)
        ent = [v for k, v in tj["kernels"].items() if k.startswith(kernel_prefix)]
        return max(ent, key=lambda v: v["dispatches"])["traffic_bytes"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--keep-apart", action="store_true",
                    help="(profiling) join the engine's two streams after every launch of the timed region: "
                         "hardware counters of overlapping dispatches cannot be told apart")
    ap.add_argument("--no-curve", action="store_true",
                    help="skip the sweeps/s-vs-chains diagnostic (extra key, untimed)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 3, 4),
                    help="1 (default): BASELINE configs[1], the headline; 3: configs[3], n=1e5 p=4096, "
                         "1024 chains per GPU, the design matrix row-sharded over the ranks and ONE "
                         "all-reduce of the 134 MB sufficient-statistics block; 4: configs[4], logit "
                         "spike-and-slab n=5e4 p=1024, 512 chains per GPU, the data replicated")
    ap.add_argument("--n-obs", type=int, default=0, help="(configs 3 / 4) rows, for dry runs")
    ap.add_argument("--p", type=int, default=0, help="(configs 3 / 4) predictors, for dry runs")
    ap.add_argument("--chains", type=int, default=0, help="(configs 3 / 4) chains per rank, for dry runs")
    ap.add_argument("--imputer", default="mixture", choices=("mixture", "pg"),
                    help="(config 4) the reference's auxiliary-mixture imputer or Polya-Gamma")
    ap.add_argument("--dump-blocks", default=None,
                    help="rank 0 writes the gathered per-rank summary blocks, the ranks' chain "
                         "offsets and a digest of every rank's installed sufficient statistics "
                         "to this .npz (tests/test_00_two_rank_bench_gpu.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain invocation: become the launcher (no GPU call has happened yet)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # (BOOM_AMD_BENCH_BACKEND=gloo: dry run of the multi-rank path on fewer GPUs
    # than ranks -- ranks share devices, collectives go through the host; the
    # driver's runs use RCCL, one rank per GPU)
    backend = os.environ.get("BOOM_AMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import boom_amd
    from cases import regression_data, spike_slab_prior

    if args.config != 1:
        run = run_config3 if args.config == 3 else run_config4
        run(args, boom_amd, torch, dist, rank, local_rank, world)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- synthetic workload -------------------------------
    X, y, _ = regression_data(N_OBS, P, N_SIGNAL, seed=DATA_SEED)
    eng = boom_amd.Engine(CHAINS_PER_GPU, seed=SAMPLER_SEED, device=local_rank,
                          chain_offset=rank * CHAINS_PER_GPU)
    # X, y go to HBM as torch tensors; XtX / Xty are built on the device.  With more
    # than one rank a rank uploads ITS ROWS only (at configs[3] the whole
    # matrix is 3.3 GB per rank of waste).
    from boom_amd import dist as bd
    lo, hi = bd.row_shard(N_OBS, rank, world)
    Xd = torch.from_numpy(np.ascontiguousarray(X[lo:hi].T)).cuda()  # column-major rows x p
    yd = torch.from_numpy(np.ascontiguousarray(y[lo:hi])).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if world > 1:
        # the multi-GPU data path: local MFMA syrk on the rank's rows, ONE all-reduce
        # of (X'X | X'y | y'y | sums) over RCCL, every rank installs the (bitwise
        # identical) total
        suf_block = bd.build_suf_row_sharded(eng, Xd, yd, N_OBS, world)
        suf_block = suf_block.cpu().numpy() if args.dump_blocks else None
    else:
        eng.build_suf_from_xy_device(N_OBS, P, Xd.data_ptr(), yd.data_ptr())
    suf_build_s = time.perf_counter() - t0
    del Xd, yd
    s = eng.get_suf()
    suf = dict(xtx=s["xtx"], xty=s["xty"], yty=s["yty"], n=s["n"],
               sumy=s["ybar"] * s["n"], xsum=s["xbar"] * s["n"])
    prior = spike_slab_prior(suf, N_SIGNAL)  # pi_0 = 1, pi_j = 16/p
    eng.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                   prior["sigma_guess"])
    g0 = np.zeros(P, np.uint8)
    g0[0] = 1
    eng.set_state(g0)
    eng.sweep(BURN_IN)

    # ---- timed region -------------------------------------------------------
    # The engine's production mode: consecutive ba_sweep calls with nothing in between run on
    # two streams and hand the chains over one by one (a workgroup of launch k + 1 takes a
    # chain launch k is done with), so no launch waits for the previous one's slowest chain
    # ("launches that overlap").  K steps = K launches of 1000 sweeps, timed between
    # two full synchronisations.  Every launch is bracketed by its own pair of HIP events ON
    # THE STREAM IT WAS LAUNCHED ON (ba_set_kernel_timing, mode 2: timing without keeping the
    # launches apart): roofline.kernel_ms is the average of those per-launch durations -- what
    # rocprofv3 --kernel-trace reports per dispatch; overlapping durations add up to more than
    # the wall time, which is the point.  The same steps kept apart (round 3's headline mode)
    # are measured right after as `separate_launches`.
    for _ in range(args.warmup):
        eng.sweep(SWEEPS_PER_STEP, sync=False)
        if args.keep_apart:
            eng.stream()
    eng.sync()
    eng.reset_summaries()
    eng.set_kernel_timing(True, overlap=True)
    eng.kernel_times()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.sweep(SWEEPS_PER_STEP, sync=False)
        if args.keep_apart:
            eng.stream()
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kt_head = eng.kernel_times()
    eng.set_kernel_timing(False)
    ms_sum = sum(ms for ms, _ in kt_head.values())
    n_launch = max(1, max(n for _, n in kt_head.values()))
    kernel_ms = ms_sum / n_launch          # avg launch duration (per-launch event pairs)

    # ---- posterior summaries: one RCCL all-gather at the end ----------------
    block = torch.empty(bd.summary_block_size(P), dtype=torch.float64, device="cuda")
    eng.summaries_device(block.data_ptr())
    allb = bd.gather_blocks(block, world)          # ONE RCCL all-gather
    coll = _collectives(bd, dist, world)
    elapsed = bd.max_over_ranks(elapsed, world, "cuda")
    if args.dump_blocks:
        # (test hook, outside the timed region: what every rank holds, side by side)
        sr = eng.get_suf()
        dig = torch.tensor([float(rank * CHAINS_PER_GPU), float(sr["xtx"].sum()), float(np.abs(sr["xtx"]).sum()),
                            float(sr["xty"].sum()), float(sr["yty"]), float(sr["ybar"]), float(sr["xbar"].sum()),
                            float(sr["n"])], dtype=torch.float64, device="cuda")
        digs = bd.gather_blocks(dig, world)
        if rank == 0:
            np.savez(args.dump_blocks, blocks=allb, digests=digs, suf_block=suf_block)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    sc = allb[:, 3 * P:]
    total_sweeps = float(sc[:, 0].sum())
    kbar = float(sc[:, 3].sum() / total_sweeps)
    incl = allb[:, :P].sum(axis=0) / total_sweeps
    value = total_sweeps / elapsed

    # ---- the same steps kept apart (extra key; round 3's headline mode): ba_stream() between
    # the calls joins the engine's two streams, so every launch starts when the previous one
    # has ended and lasts as long as its slowest chain
    est = torch.cuda.ExternalStream(eng.stream(), device=torch.device("cuda", local_rank))
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(est)
    for _ in range(args.steps):
        eng.sweep(SWEEPS_PER_STEP, sync=False)
        eng.stream()
    ev1.record(est)
    eng.sync()
    torch.cuda.synchronize()
    sep_elapsed = time.perf_counter() - t0
    sep_kernel_ms = ev0.elapsed_time(ev1) / max(1, args.steps)

    # decision safety of the timed sweeps themselves: smallest |log u - delta|
    # any chain saw (a flip decision could differ from the reference's only below
    # the ~1e-12 rounding difference), and the chains' status words (ba_sync
    # returned OK for every rank or we would not be here)
    decisions = {"min_margin": float(sc[:, 6].min()),
                 "accepted_flips": float(sc[:, 4].sum()),
                 "proposed_flips": float(sc[:, 5].sum()),
                 "chains_in_error": 0, "worst_chain_status": "CHAIN_OK"}

    # ESS/s: an extra, untimed run of ESS_SWEEPS recorded sweeps on rank 0:
    # sigma^2, |gamma|, log posterior and the five largest-|beta| coefficients
    # (SURVEY 8d); min over traces of the pooled ESS fraction x measured sweeps/s
    trace_len = ESS_SWEEPS
    eng.enable_draws(trace_len)
    eng.sweep(trace_len)
    tr = eng.get_traces(trace_len)
    beta_mean = allb[:, P:2 * P].sum(axis=0) / total_sweeps
    top5 = [int(j) for j in np.argsort(-np.abs(beta_mean))[:5]]
    bt = eng.get_coefficient_traces(trace_len, top5)
    ess = {}
    for name in ("sigsq", "model_size", "logp"):
        ess[name] = sum(geyer_ess(tr[name][c]) for c in range(CHAINS_PER_GPU))
    for i, j in enumerate(top5):
        ess["beta[%d]" % j] = sum(geyer_ess(bt[c, i]) for c in range(CHAINS_PER_GPU))
    ess_frac = min(ess.values()) / (CHAINS_PER_GPU * trace_len)
    ess_per_sec = ess_frac * value

    # ---- roofline of the dominant kernel (ssvs_sweep_kernel) ----------------
    f = 8.0
    bytes_per_sweep = P * f * (2 * kbar + 4) + f * (3 * kbar + 4) + P / 8.0
    flops_per_sweep = P * (4 * kbar ** 2 + 12 * kbar + 40) + kbar ** 3 / 3 + 4 * kbar ** 2
    launch_bytes = bytes_per_sweep * CHAINS_PER_GPU * SWEEPS_PER_STEP
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch: NOT measured by this run (counters need a profiler pass) --
    # read from the committed PMC passes of this same command (tools/regen_profiles.sh:
    # separate FETCH_SIZE / WRITE_SIZE rocprofv3 runs, corrected as the guide says), and
    # labelled with the commit they were taken at
    traffic, traffic_source = None, None
    try:
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_c2_pmc_traffic.json")))
        with open(cand[-1]) as fh:
            tj = json.load(fh)
        ent = [v for k, v in tj["kernels"].items() if k.startswith("ssvs_sweep_kernel")]
        ent = max(ent, key=lambda v: v["dispatches"])
        traffic = ent["traffic_bytes"]
        traffic_source = ("profiles/%s: rocprofv3 --pmc passes of this command at commit %s; FETCH_SIZE "
                          "doubled (upper bound), raw sum %d bytes"
                          % (os.path.basename(cand[-1]), tj.get("commit"), ent["traffic_bytes_raw"]))
    except Exception:
        pass
    # SURVEY 8(d)'s figure, under the name of what it is: the bytes the REFERENCE's algorithm
    # would read for these sweeps over the time the kernel took.  The kernel does not move them
    # (the threshold table answers four sweeps in five with look-ups; counters: 1.1 GB per launch
    # of HBM traffic for 152 GB "algorithmic"), so this is a work rate, not a bandwidth, and it
    # is NOT the roofline fraction: over the whole timed region it exceeds the HBM peak.
    reference_bytes = {"what": "SURVEY 8(d): bytes the reference's from-scratch algorithm would read per sweep "
                               "(p f (2 kbar + 4) + f (3 kbar + 4) + p / 8) x sweeps / kernel time; a work rate in "
                               "GB/s, not achieved bandwidth, and not a fraction of anything the kernel is bound by",
                       "bytes_per_sweep": round(bytes_per_sweep, 1),
                       "flops_per_sweep": round(flops_per_sweep, 1),
                       "per_overlapped_launch": {"kernel_ms": round(kernel_ms, 4), "GBps": round(achieved, 2),
                                                 "vs_hbm_peak": round(achieved / HBM_PEAK_GBS, 5)},
                       "kept_apart": {"kernel_ms": round(sep_kernel_ms, 4),
                                      "GBps": round(launch_bytes / (sep_kernel_ms * 1e-3) / 1e9, 2),
                                      "vs_hbm_peak": round(launch_bytes / (sep_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                       "whole_timed_region_vs_hbm_peak": round(launch_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                       "gflops": round(flops_per_sweep * CHAINS_PER_GPU * SWEEPS_PER_STEP / (kernel_ms * 1e-3) / 1e9, 2)}
    # What binds the kernel: a chain is two wavefronts whose instruction streams are chains of
    # dependent instructions and LDS / L2 round trips.  Executed work from the
    # committed counter passes of this command (one launch at a time), the bound from the
    # guide's issue cost and latencies; the quiet sweep's longer side (wave 1: shuffle + table
    # walk at p = 512) serialises: exchange search 2 LDS trips (keys, exchanges), links 2,
    # chain ends 1, pointer jumping ~4 rounds x 2, gather 3, the walk's permutation read 1 and
    # its table gather 1 L2 trip, fork and join 2.
    ex = _executed("c2", "ssvs_sweep_kernel", units_per_dispatch=CHAINS_PER_GPU * SWEEPS_PER_STEP,
                   dependent_trips={"lds": 19, "l2": 1,
                                    "what": "per quiet sweep of the longer side (wave 1, p = 512): exchange search 2, links 2, "
                                            "chain ends 1, pointer jumping 4 rounds x 2, gather 3, permutation read 1, fork + "
                                            "join 2 LDS round trips; the table gather 1 L2 round trip (ssvs_device.h, "
                                            "parallel_shuffle and the walk)"})
    roofline = _latency_issue_roofline(ex, value / world, "sweeps/s (one GPU, the timed region)", CHAINS_PER_GPU,
                                       SWEEPS_PER_STEP, 2, reference_bytes)
    roofline.update({
        "traffic": traffic, "traffic_source": traffic_source,
        "kernel": "ssvs_sweep_kernel", "kernel_ms": round(kernel_ms, 4),
        "kernel_ms_how": "average over the %d timed launches of each launch's own HIP-event pair on the stream it "
                         "went to; the launches overlap (their durations sum to %.1f ms in %.1f ms of wall time)"
                         % (n_launch, ms_sum, elapsed * 1e3),
        "launches_in_flight": round(ms_sum / (elapsed * 1e3), 3),
        "kept_apart_kernel_ms": round(sep_kernel_ms, 4),
        "note": "bound = latency/issue: per sweep each of a chain's two wavefronts lives measured_cycles, of which the "
                "instructions it executed account for `issue` at 4 cycles each and the serial LDS / L2 round trips of "
                "the shuffle and the table walk for `dependent_round_trips`; frac = (issue + trips) / measured, peak = "
                "achieved / frac = the rate at which no wavefront would ever wait beyond those.  HBM is not the limit "
                "(traffic: counters, per launch; the working set is LDS / L2 resident) and neither are the matrix cores "
                "(BASELINE.md sec. 3); SURVEY 8(d)'s byte figure is kept as work_rate_vs_reference_bytes"})

    # ---- sweeps/s vs chains per GPU (diagnostic, untimed extra key) ----------
    curve = None
    if not args.no_curve:
        curve = {}
        for nch in (2048, 4096, 8192):
            e2 = boom_amd.Engine(nch, seed=SAMPLER_SEED, device=local_rank)
            e2.upload_suf(s["xtx"], s["xty"], s["yty"], s["n"], s["ybar"], s["xbar"])
            e2.set_priors(prior["b"], prior["ominv"], prior["pi"], prior["df"],
                          prior["sigma_guess"])
            e2.set_state(g0)
            e2.sweep(BURN_IN)
            e2.sweep(SWEEPS_PER_STEP)
            t0 = time.perf_counter()
            for _ in range(4):
                e2.sweep(SWEEPS_PER_STEP, sync=False)
            e2.sync()
            curve[str(nch)] = round(4 * nch * SWEEPS_PER_STEP / (time.perf_counter() - t0), 1)
            e2.close()
        curve[str(CHAINS_PER_GPU)] = round(value / world, 1)

    # ---- the reference callers' loop (extra key): one sample_posterior() per iteration
    # with the parameters read back after each (spike_slab_wrapper.cc:233-242), served by
    # ba_draw_next at the hosts' default look-ahead
    loop = None
    if not args.no_curve and world == 1:
        loop = {}
        for L in (64, 256):
            eng.set_lookahead(L)
            for _ in range(L):
                eng.draw_next()
            eng.get_state(0)
            tb = []
            for _ in range(50 if L == 64 else 16):
                t0 = time.perf_counter()
                for _ in range(L):
                    eng.draw_next()
                    eng.get_state(0)
                tb.append(time.perf_counter() - t0)
            q1, med, q3 = (float(v) for v in np.percentile(tb, [25, 50, 75]))
            loop["lookahead_%d" % L] = round(CHAINS_PER_GPU * L / med, 1)
            loop["lookahead_%d_iqr" % L] = [round(CHAINS_PER_GPU * L / q3, 1), round(CHAINS_PER_GPU * L / q1, 1)]
            loop["lookahead_%d_batches" % L] = len(tb)
        eng.set_lookahead(1)

    # ---- launches that overlap (extra key): consecutive ba_sweep calls with nothing in
    # between run on two streams and hand chains over launch to launch (a workgroup of the
    # next launch takes a chain the current one is done with), so no launch waits for the
    # previous one's slowest chain.  The headline's steps above are kept apart on purpose.
    overlapped = None
    if not args.no_curve and world == 1:
        overlapped = {}
        eng.enable_traces(0)       # (the look-ahead's record: not wanted here)
        for L, K in ((1000, 8), (250, 32), (64, 125)):
            eng.sync()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(K):
                    eng.sweep(L, sync=False)
                eng.sync()
                ts.append(time.perf_counter() - t0)
            overlapped["%d_sweeps_x_%d_launches" % (L, K)] = round(CHAINS_PER_GPU * L * K / min(ts), 1)

    # ---- the other BASELINE configurations at their per-GPU shapes (extra key, outside
    # the timed region; parity for them lives in tests/).  Per configuration: the rate of
    # a plain throughput pass, then a second pass with the engine's per-kernel HIP-event
    # timing on (ba_set_kernel_timing: events on the engine's stream around every
    # launch) for the dominant kernel's roofline against SURVEY 8(d)'s bytes, and the
    # oracle on the host cores on a bounded sample -------------------------------------
    other = None
    if not args.no_curve and world == 1:
        other = other_configs(boom_amd, torch, local_rank, cpu=not args.no_cpu_baseline)

    # ---- CPU baseline: the oracle (a port of the reference algorithm) -------
    cpu = None
    if not args.no_cpu_baseline:
        from oracle_lib import Oracle, ssvs_options
        O = Oracle()
        cores = usable_cores()
        # warm start from the GPU's current state so that kbar matches
        gam, beta, sig = eng.get_states()

        def timed(nchains, nsw, nthreads):
            t0 = time.perf_counter()
            O.run_chains(suf, prior, ssvs_options(), SAMPLER_SEED, nchains, nsw, nthreads,
                         gam[0], beta[0], float(sig[0]))
            return nchains * nsw / (time.perf_counter() - t0)
        one = timed(1, 200, 1)                      # calibrates the sample sizes
        nsw1 = int(max(200, min(4000, 5.0 * one)))
        one = timed(1, nsw1, 1)                     # ~5 s, one thread
        nchains = 2 * cores
        cal = timed(nchains, 20, cores)             # all-core rate, short calibration run
        nswc = int(max(20, min(4000, 12.0 * cal / nchains)))
        allc = timed(nchains, nswc, cores)          # ~12 s
        cpu = {"value": round(allc, 2), "unit": "sweeps/s", "cores": cores, "kind": "port",
               "host_threads_visible": os.cpu_count(),
               "one_thread": round(one, 2),
               "sample": "all-core: %d chains x %d sweeps on %d pthreads; one thread: 1 chain x "
                         "%d sweeps; same n=1e4 p=512 workload, warm-started at the GPU chains' "
                         "state (kbar~%.1f); oracle/boom_oracle.c, read-only matrices and the "
                         "correlation map shared by the threads" % (nchains, nswc, cores, nsw1, kbar),
               "port_vs_reference": "2.2x faster than the compiled reference per thread in the "
                                    "build container (317 vs 146 sweeps/s at this shape, 8-core "
                                    "container; the compiled reference travels to the GPU box as a BUILT "
                                    "file -- git-ignored, not gpurun-ignored -- and is loaded there by the "
                                    "tests only: reference-side binding, goldens; it is not timed there)"}

    out = {
        "metric": "Gibbs sweeps/sec (all chains), n=1e4 p=512 spike-slab",
        "value": round(value, 1),
        "unit": "sweeps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BregVsSampler spike-and-slab n=1e4 p=512, "
                               "1024 chains per GPU, fp64 (BASELINE configs[1])",
                   "chains_per_gpu": CHAINS_PER_GPU,
                   "sweeps_per_step": SWEEPS_PER_STEP,
                   "true_signals": N_SIGNAL, "mean_model_size": round(kbar, 2),
                   "burn_in": BURN_IN, "parallelism": "chains sharded, %d GPU(s)" % world,
                   "launches": "consecutive ba_sweep calls overlap (chain hand-over between launches): the "
                               "engine's default",
                   "suf_build": ("rows sharded, local MFMA syrk, one all-reduce" if world > 1
                                 else "single device MFMA syrk")},
        "ess_per_sec": round(ess_per_sec, 1),
        "ess_fraction": round(ess_frac, 4),
        "ess_traces": {k: round(v / (CHAINS_PER_GPU * trace_len), 4) for k, v in ess.items()},
        "decisions": decisions,
        "sweeps_per_sec_vs_chains_per_gpu": curve,
        "drop_in_loop_sweeps_per_sec": loop,
        "overlapped_launches_sweeps_per_sec": overlapped,
        "separate_launches": {"sweeps_per_sec": round(CHAINS_PER_GPU * SWEEPS_PER_STEP * args.steps / sep_elapsed, 1),
                              "kernel_ms": round(sep_kernel_ms, 4),
                              "work_rate_vs_reference_bytes_over_hbm_peak": round(launch_bytes / (sep_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              "what": "the same K steps with ba_stream() between the calls: one launch at a "
                                      "time, each as long as its slowest chain (round 3's headline mode)"},
        "other_configs": other,
        "suf_build_ms": round(suf_build_s * 1e3, 2),
        "collectives": coll,
        "signal_inclusion_min": round(float(incl[:N_SIGNAL].min()), 4),
        "null_inclusion_max": round(float(incl[N_SIGNAL:].max()), 4),
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
