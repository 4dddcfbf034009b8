"""ctypes plumbing over the C-ABI (include/boom_amd.h, boom_amd/libboomamd.so).

This is NOT a compute path: every call goes straight into the HIP library and
fails loudly when the library or a GPU is missing.  There is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BOOM_AMD_LIB selects another build of the SAME library (e.g. the -DBA_STAMPS
# diagnostic build); it is never a different backend
LIB_PATH = os.environ.get("BOOM_AMD_LIB", os.path.join(_HERE, "libboomamd.so"))

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)

SUMMARY_SCALARS = 16


class BoomAmdError(RuntimeError):
    """The C-ABI returned an error; mirrors BOOM's report_error ->
    std::runtime_error (cpputil/report_error.cpp:30-32)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class BaConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("chains", C.c_int32),
                ("chain_offset", C.c_int64), ("seed", C.c_uint64),
                ("max_model_size_hint", C.c_int32), ("reserved", C.c_int32)]


_lib = None

# name -> (restype, argtypes); must list every symbol include/boom_amd.h declares
SIGNATURES = {
    "ba_last_error": (C.c_char_p, []),
    "ba_engine_create": (C.c_int, [C.POINTER(BaConfig), C.POINTER(C.c_void_p)]),
    "ba_engine_destroy": (None, [C.c_void_p]),
    "ba_engine_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ba_build_suf_from_xy": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _dp, _dp]),
    "ba_build_suf_from_xy_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32,
                                              C.c_void_p, C.c_void_p]),
    "ba_suf_block_size": (C.c_size_t, [C.c_int32]),
    "ba_suf_partial_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "ba_set_suf_from_block_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "ba_upload_regression_suf": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp,
                                           C.c_double, C.c_double, C.c_double, _dp]),
    "ba_get_regression_suf": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]),
    "ba_set_slab": (C.c_int, [C.c_void_p, _dp, _dp]),
    "ba_set_spike": (C.c_int, [C.c_void_p, _dp, C.c_int64]),
    "ba_set_sigma_prior": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "ba_set_priors_ctor1": (C.c_int, [C.c_void_p, C.c_double, C.c_double,
                                      C.c_double, C.c_int32]),
    "ba_set_priors_ctor2": (C.c_int, [C.c_void_p, C.c_double, C.c_double,
                                      C.c_double, C.c_double, C.c_double, C.c_int32]),
    "ba_get_priors": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp]),
    "ba_set_options": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32]),
    "ba_set_tuning": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "ba_set_lookahead": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_draw_next": (C.c_int, [C.c_void_p]),
    "ba_set_state": (C.c_int, [C.c_void_p, C.c_int64, _u8p, _dp, C.c_double]),
    "ba_get_state": (C.c_int, [C.c_void_p, C.c_int64, _u8p, _dp, _dp]),
    "ba_get_states": (C.c_int, [C.c_void_p, _u8p, _dp, _dp]),
    "ba_logpri": (C.c_int, [C.c_void_p, C.c_int64, _dp]),
    "ba_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "ba_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_sync": (C.c_int, [C.c_void_p]),
    "ba_log_model_prob": (C.c_int, [C.c_void_p, C.c_int32, _u8p, _dp]),
    "ba_set_sigsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_double]),
    "ba_sss_set_slab": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32, C.c_int32]),
    "ba_sss_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_adaptive_set_options": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double]),
    "ba_adaptive_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_adaptive_get_rates": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp,
                                        C.POINTER(C.c_uint64)]),
    "ba_reset_summaries": (C.c_int, [C.c_void_p]),
    "ba_get_summaries": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "ba_summaries_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ba_enable_traces": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_get_traces": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp]),
    "ba_enable_draws": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_get_draws": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _u8p, _dp, _dp]),
    "ba_predict": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]),
    "ba_get_coefficient_traces": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32,
                                            C.POINTER(C.c_int32), _dp]),
    "ba_stream": (C.c_void_p, [C.c_void_p]),
    "ba_kernel_classes": (C.c_int32, []),
    "ba_kernel_class_name": (C.c_char_p, [C.c_int32]),
    "ba_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_get_kernel_times": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int64), C.c_int32]),
    "ba_group_last_error": (C.c_char_p, []),
    "ba_group_rccl_available": (C.c_int32, []),
    "ba_group_create": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_uint64,
                                  C.POINTER(C.c_void_p)]),
    "ba_group_destroy": (None, [C.c_void_p]),
    "ba_group_size": (C.c_int32, [C.c_void_p]),
    "ba_group_engine": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "ba_group_locate": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ba_group_build_suf_from_xy": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _dp, _dp]),
    "ba_group_set_priors": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int64, C.c_double, C.c_double,
                                      C.c_double]),
    "ba_group_set_state": (C.c_int, [C.c_void_p, _u8p, _dp, C.c_double]),
    "ba_group_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_group_sync": (C.c_int, [C.c_void_p]),
    "ba_group_call": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ba_group_ss_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_group_ss_draw_next": (C.c_int, [C.c_void_p]),
    "ba_group_logit_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_group_poisson_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_group_reset_summaries": (C.c_int, [C.c_void_p]),
    "ba_group_get_summaries": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp]),
    "ba_ss_set_data": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _u8p]),
    "ba_ss_set_local_level": (C.c_int, [C.c_void_p] + [C.c_double] * 6),
    "ba_probit_set_data": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _dp, _dp, _dp, C.c_int32]),
    "ba_probit_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_logit_set_data": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _dp, _dp, _dp, C.c_int32]),
    "ba_logit_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_logit_set_imputer": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_poisson_set_data": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _dp, _dp, _dp]),
    "ba_poisson_set_mixtures": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int32), _dp, _dp, _dp, C.c_int64]),
    "ba_poisson_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_ss_set_structural": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32] + [_dp] * 6),
    "ba_ss_get_structural": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp]),
    "ba_ss_add_ar": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double,
                               C.c_double, _dp, _dp, _dp]),
    "ba_ss_get_ar": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp]),
    "ba_ss_clear_state_models": (C.c_int, [C.c_void_p]),
    "ba_ss_set_tuning": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_set_slot_limit": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_ss_add_state_model": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)] + [_dp] * 7),
    "ba_ss_state_dimension": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ba_ss_get_state_model": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32] + [_dp] * 8),
    "ba_ss_get_state_draw": (C.c_int, [C.c_void_p, C.c_int64, _dp]),
    "ba_ss_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_ss_set_lookahead": (C.c_int, [C.c_void_p, C.c_int32]),
    "ba_ss_lookahead_chains": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "ba_ss_draw_next": (C.c_int, [C.c_void_p]),
    "ba_ss_impute_state": (C.c_int, [C.c_void_p]),
    "ba_ss_forecast": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
    "ba_ss_get_state": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp]),
    "ba_ss_set_level_sigsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_double]),
    "ba_ss_get_chain_suf": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp, _dp]),
}


def load_library():
    """Load libboomamd.so (no GPU needed just to load it)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "boom_amd/libboomamd.so is missing: run `python -c 'import "
                "__graft_entry__ as g; g.build()'` (hipcc, gfx950).  There is "
                "no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _fcol(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).ravel()


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _b(a):
    return None if a is None else a.ctypes.data_as(_u8p)


class Engine:
    """One engine = `chains` independent chains on one HIP device."""

    def __init__(self, chains, seed=8675309, device=0, chain_offset=0,
                 max_model_size_hint=0, _borrowed=None):
        self.lib = load_library()
        self._owned = _borrowed is None
        self._h = None
        if _borrowed is None:
            cfg = BaConfig(device, chains, chain_offset, seed, max_model_size_hint, 0)
            h = C.c_void_p()
            self._check(self.lib.ba_engine_create(C.byref(cfg), C.byref(h)))
            self._h = h
        else:
            self._h = C.c_void_p(_borrowed)   # an engine of a Group: the group owns it
        self.chains = chains
        self.p = 0

    def _check(self, rc):
        if rc != 0:
            raise BoomAmdError(rc, self.lib.ba_last_error().decode())

    def close(self):
        if self._h is not None:
            if self._owned:
                self.lib.ba_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data ---------------------------------------------------------------
    def build_suf_from_xy(self, X, y):
        n, p = X.shape
        self._check(self.lib.ba_build_suf_from_xy(self._h, n, p, _p(_fcol(X)), _p(_f64(y))))
        self.p = p

    def build_suf_from_xy_device(self, n, p, x_ptr, y_ptr):
        self._check(self.lib.ba_build_suf_from_xy_device(self._h, n, p, x_ptr, y_ptr))
        self.p = p

    def suf_block_size(self, p):
        return int(self.lib.ba_suf_block_size(p))

    def suf_partial_device(self, n_rows, p, x_ptr, y_ptr, block_ptr):
        self._check(self.lib.ba_suf_partial_device(self._h, n_rows, p, x_ptr, y_ptr, block_ptr))

    def set_suf_from_block_device(self, n_total, p, block_ptr):
        self._check(self.lib.ba_set_suf_from_block_device(self._h, n_total, p, block_ptr))
        self.p = p

    def upload_suf(self, xtx, xty, yty, n, ybar, xbar):
        p = len(xty)
        self._check(self.lib.ba_upload_regression_suf(
            self._h, p, _p(_fcol(xtx)), _p(_f64(xty)), yty, n, ybar, _p(_f64(xbar))))
        self.p = p

    def get_suf(self):
        p = self.p
        xtx = np.zeros(p * p)
        xty = np.zeros(p)
        xbar = np.zeros(p)
        yty, n, ybar = C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.ba_get_regression_suf(self._h, _p(xtx), _p(xty), C.byref(yty),
                                                   C.byref(n), C.byref(ybar), _p(xbar)))
        return dict(xtx=xtx.reshape(p, p).T.copy(), xty=xty, yty=yty.value,
                    n=n.value, ybar=ybar.value, xbar=xbar)

    # ---- priors -------------------------------------------------------------
    def set_priors(self, b, ominv, pi, prior_df, sigma_guess,
                   max_model_size=-1, sigma_upper_limit=float("inf")):
        self._check(self.lib.ba_set_slab(self._h, _p(_f64(b)), _p(_fcol(ominv))))
        self._check(self.lib.ba_set_spike(self._h, _p(_f64(pi)), int(max_model_size)))
        self._check(self.lib.ba_set_sigma_prior(self._h, prior_df, sigma_guess,
                                                sigma_upper_limit))

    def set_priors_ctor1(self, prior_nobs, expected_rsq, expected_model_size,
                         first_term_is_intercept=True):
        self._check(self.lib.ba_set_priors_ctor1(self._h, prior_nobs, expected_rsq,
                                                 expected_model_size,
                                                 int(first_term_is_intercept)))

    def set_priors_ctor2(self, prior_sigma_nobs, prior_sigma_guess, prior_beta_nobs,
                         diagonal_shrinkage, prior_inclusion_probability,
                         force_intercept=True):
        self._check(self.lib.ba_set_priors_ctor2(
            self._h, prior_sigma_nobs, prior_sigma_guess, prior_beta_nobs,
            diagonal_shrinkage, prior_inclusion_probability, int(force_intercept)))

    def get_priors(self):
        p = self.p
        b, om, pi = np.zeros(p), np.zeros(p * p), np.zeros(p)
        df, ss = C.c_double(), C.c_double()
        self._check(self.lib.ba_get_priors(self._h, _p(b), _p(om), _p(pi),
                                           C.byref(df), C.byref(ss)))
        return dict(b=b, ominv=om.reshape(p, p).T.copy(), pi=pi, df=df.value, ss=ss.value)

    def set_options(self, max_flips=-1, swap_threshold=0.8, draw_beta=True,
                    draw_sigma=True):
        self._check(self.lib.ba_set_options(self._h, max_flips, swap_threshold,
                                            int(draw_beta), int(draw_sigma)))

    def set_tuning(self, waves_per_chain=0, walk_policy=-1, kcap_start=0):
        self._check(self.lib.ba_set_tuning(self._h, waves_per_chain, walk_policy, kcap_start))

    # ---- state --------------------------------------------------------------
    def set_state(self, gamma, beta=None, sigsq=1.0, chain=-1):
        g = np.ascontiguousarray(gamma, dtype=np.uint8)
        bb = None if beta is None else _f64(beta)
        self._check(self.lib.ba_set_state(self._h, chain, _b(g), _p(bb), sigsq))

    def get_state(self, chain):
        p = self.p
        g = np.zeros(p, np.uint8)
        b = np.zeros(p)
        s = C.c_double()
        self._check(self.lib.ba_get_state(self._h, chain, _b(g), _p(b), C.byref(s)))
        return g, b, s.value

    def get_states(self):
        p, c = self.p, self.chains
        g = np.zeros((c, p), np.uint8)
        b = np.zeros((c, p))
        s = np.zeros(c)
        self._check(self.lib.ba_get_states(self._h, _b(g), _p(b), _p(s)))
        return g, b, s

    def seed(self, seed):
        self._check(self.lib.ba_seed(self._h, seed))

    # ---- sampling -----------------------------------------------------------
    def sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def sync(self):
        self._check(self.lib.ba_sync(self._h))

    def set_lookahead(self, n):
        self._check(self.lib.ba_set_lookahead(self._h, n))

    def draw_next(self):
        self._check(self.lib.ba_draw_next(self._h))

    def log_model_prob(self, gammas):
        G = np.ascontiguousarray(gammas, dtype=np.uint8)
        out = np.zeros(len(G))
        self._check(self.lib.ba_log_model_prob(self._h, len(G), _b(G), _p(out)))
        return out

    # ---- SpikeSlabSampler (sigma^2 given) --------------------------------------
    def set_sigsq(self, sigsq, chain=-1):
        self._check(self.lib.ba_set_sigsq(self._h, chain, float(sigsq)))

    def sss_set_slab(self, mu, precision, scales_with_sigsq=True, max_flips=-1):
        self._check(self.lib.ba_sss_set_slab(self._h, _p(_f64(mu)), _p(_fcol(precision)),
                                             int(scales_with_sigsq), int(max_flips)))

    def set_spike(self, pi, max_model_size=-1):
        self._check(self.lib.ba_set_spike(self._h, _p(_f64(pi)), int(max_model_size)))

    def sss_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_sss_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    # ---- AdaptiveSpikeSlabRegressionSampler ---------------------------------------
    def adaptive_set_options(self, max_flips=-1, step_size=-1.0, target=-1.0):
        self._check(self.lib.ba_adaptive_set_options(self._h, max_flips, step_size, target))

    def adaptive_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_adaptive_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def adaptive_get_rates(self, chain):
        b, d = np.zeros(self.p), np.zeros(self.p)
        it = C.c_uint64()
        self._check(self.lib.ba_adaptive_get_rates(self._h, chain, _p(b), _p(d), C.byref(it)))
        return b, d, it.value

    # ---- summaries ----------------------------------------------------------
    def reset_summaries(self):
        self._check(self.lib.ba_reset_summaries(self._h))

    def get_summaries(self):
        p = self.p
        inc, bs, bs2 = np.zeros(p), np.zeros(p), np.zeros(p)
        sc = np.zeros(SUMMARY_SCALARS)
        self._check(self.lib.ba_get_summaries(self._h, _p(inc), _p(bs), _p(bs2), _p(sc)))
        return dict(inclusion_count=inc, beta_sum=bs, beta_sumsq=bs2,
                    sweeps=sc[0], sigsq_sum=sc[1], sigsq_sumsq=sc[2], k_sum=sc[3],
                    accepts=sc[4], proposals=sc[5], min_margin=sc[6],
                    # scalar 7: accepted flips served from a chain's other slot
                    # (diagnostic stamp builds reuse it for the slowest chain's cycles)
                    slot_hits=float(sc[7]), phase_cycles=sc[8:16].copy(),
                    # after adaptive sweeps: closest approach of a weighted-draw
                    # uniform to a boundary of the cumulative rates
                    min_multi_margin=float(sc[8]))

    def summaries_device(self, ptr):
        self._check(self.lib.ba_summaries_device(self._h, ptr))

    def enable_traces(self, max_sweeps):
        self._check(self.lib.ba_enable_traces(self._h, max_sweeps))

    def logpri(self, chain=0):
        out = C.c_double()
        self._check(self.lib.ba_logpri(self._h, chain, C.byref(out)))
        return out.value

    def enable_draws(self, max_sweeps):
        self._check(self.lib.ba_enable_draws(self._h, max_sweeps))

    def get_draws(self, chain, nsweeps):
        """the recorded draws of one chain (local index) of the last sweep() call"""
        p = self.p
        g = np.zeros((nsweeps, p), np.uint8)
        b = np.zeros((nsweeps, p))
        s = np.zeros(nsweeps)
        self._check(self.lib.ba_get_draws(self._h, chain, nsweeps, _b(g), _p(b), _p(s)))
        return g, b, s

    def predict(self, newX, first_draw, ndraws):
        """posterior predictive means newX beta for the recorded draws [first_draw,
        first_draw + ndraws) of every chain: chains x ndraws x len(newX)"""
        Xn = np.asfortranarray(np.atleast_2d(newX), dtype=np.float64)
        out = np.zeros((self.chains, ndraws, Xn.shape[0]))
        self._check(self.lib.ba_predict(self._h, first_draw, ndraws, Xn.shape[0],
                                        Xn.ctypes.data_as(_dp), _p(out)))
        return out

    def get_coefficient_traces(self, nsweeps, variables):
        v = np.ascontiguousarray(variables, dtype=np.int32)
        out = np.zeros((self.chains, len(v), nsweeps))
        self._check(self.lib.ba_get_coefficient_traces(
            self._h, nsweeps, len(v), v.ctypes.data_as(C.POINTER(C.c_int32)), _p(out)))
        return out

    def get_traces(self, nsweeps):
        c = self.chains
        s, l, k = (np.zeros((c, nsweeps)) for _ in range(3))
        self._check(self.lib.ba_get_traces(self._h, nsweeps, _p(s), _p(l), _p(k)))
        return dict(sigsq=s, logp=l, model_size=k)

    def stream(self):
        return self.lib.ba_stream(self._h)

    # ---- measurement ----------------------------------------------------------
    def set_kernel_timing(self, enabled=True, overlap=False):
        """overlap=True: consecutive sweep launches still overlap while timed (mode 2)"""
        self._check(self.lib.ba_set_kernel_timing(self._h, 2 if (enabled and overlap) else int(enabled)))

    def kernel_times(self, reset=True):
        """{kernel class: (milliseconds, launches)} since the last reset"""
        n = self.lib.ba_kernel_classes()
        ms = np.zeros(n)
        cnt = np.zeros(n, np.int64)
        self._check(self.lib.ba_get_kernel_times(
            self._h, _p(ms), cnt.ctypes.data_as(C.POINTER(C.c_int64)), int(reset)))
        return {self.lib.ba_kernel_class_name(i).decode(): (float(ms[i]), int(cnt[i]))
                for i in range(n) if cnt[i] > 0}

    # ---- state space --------------------------------------------------------
    def ss_set_data(self, y, X, observed=None):
        T, p = X.shape
        obs = None if observed is None else np.ascontiguousarray(observed, np.uint8)
        self._check(self.lib.ba_ss_set_data(self._h, T, p, _p(_f64(y)), _p(_fcol(X)), _b(obs)))
        self.p = p
        self.T = T

    def ss_set_local_level(self, level_df, level_sigma_guess, level_sigma_upper_limit,
                           initial_state_mean, initial_state_variance,
                           initial_level_sigma):
        self._check(self.lib.ba_ss_set_local_level(
            self._h, level_df, level_sigma_guess, level_sigma_upper_limit,
            initial_state_mean, initial_state_variance, initial_level_sigma))

    def probit_set_data(self, X, y, ntrials, clt_threshold=5):
        X = np.asfortranarray(X, dtype=np.float64)
        self.p = X.shape[1]
        y = np.ascontiguousarray(y, dtype=np.float64)
        nt = np.ascontiguousarray(ntrials, dtype=np.float64)
        self._check(self.lib.ba_probit_set_data(self._h, X.shape[0], X.shape[1], _p(X), _p(y),
                                                _p(nt), int(clt_threshold)))

    def probit_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_probit_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def logit_set_data(self, X, y, ntrials, clt_threshold=5):
        X = np.asfortranarray(X, dtype=np.float64)
        self.p = X.shape[1]
        y = np.ascontiguousarray(y, dtype=np.float64)
        nt = np.ascontiguousarray(ntrials, dtype=np.float64)
        self._check(self.lib.ba_logit_set_data(self._h, X.shape[0], X.shape[1], _p(X), _p(y),
                                               _p(nt), int(clt_threshold)))

    def poisson_set_data(self, X, y, exposure, mix):
        """mix: dict(counts, ncomp, mu, sigma, weight, largest_index) -- the reference
        table's mixtures for 1 and the positive counts in y"""
        X = np.asfortranarray(X, dtype=np.float64)
        self.p = X.shape[1]
        self._check(self.lib.ba_poisson_set_data(self._h, X.shape[0], X.shape[1], _p(X), _p(_f64(y)),
                                                 _p(_f64(exposure))))
        counts = np.ascontiguousarray(mix["counts"], dtype=np.int64)
        ncomp = np.ascontiguousarray(mix["ncomp"], dtype=np.int32)
        self._check(self.lib.ba_poisson_set_mixtures(
            self._h, len(counts), counts.ctypes.data_as(C.POINTER(C.c_int64)),
            ncomp.ctypes.data_as(C.POINTER(C.c_int32)), _p(_f64(mix["mu"])), _p(_f64(mix["sigma"])),
            _p(_f64(mix["weight"])), int(mix["largest_index"])))

    def poisson_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_poisson_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def logit_set_imputer(self, kind):
        """0: the reference's auxiliary mixture (default); 1: Polya-Gamma"""
        self._check(self.lib.ba_logit_set_imputer(self._h, int(kind)))

    def logit_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_logit_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def ss_set_structural(self, trend, nseasons, var_df, var_sigma_guess,
                          var_sigma_upper_limit, var_initial_sigma, initial_state_mean,
                          initial_state_variance):
        """trend (1 local level, 2 local linear trend) + optional seasonal state"""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in
                (var_df, var_sigma_guess, var_sigma_upper_limit, var_initial_sigma,
                 initial_state_mean, initial_state_variance)]
        self._ssm_dim = int(trend) + (int(nseasons) - 1 if nseasons > 0 else 0)
        self._ar_lags = 0
        self._check(self.lib.ba_ss_set_structural(self._h, int(trend), int(nseasons),
                                                  *[_p(a) for a in arrs]))

    def ss_add_ar(self, lags, df, sigma_guess, sigma_upper_limit, initial_sigma, initial_phi,
                  initial_state_mean, initial_state_variance):
        """appends an ArStateModel(lags) block (+ ArPosteriorSampler) to the structural state"""
        ph = (None if initial_phi is None
              else np.ascontiguousarray(initial_phi, dtype=np.float64))
        a0 = np.ascontiguousarray(initial_state_mean, dtype=np.float64)
        p0 = np.ascontiguousarray(initial_state_variance, dtype=np.float64)
        if a0.shape != (lags,) or p0.shape != (lags,) or (ph is not None and ph.shape != (lags,)):
            raise ValueError("the block's initial moments and coefficients have `lags` entries")
        self._check(self.lib.ba_ss_add_ar(self._h, int(lags), float(df), float(sigma_guess),
                                          float(sigma_upper_limit), float(initial_sigma),
                                          _p(ph) if ph is not None else None, _p(a0), _p(p0)))
        self._ssm_dim += int(lags)
        self._ar_lags = int(lags)

    def ss_get_ar(self, chain):
        L = self._ar_lags
        phi, xtx, xty = np.zeros(L), np.zeros((L, L)), np.zeros(L)
        sig, yty, n = C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.ba_ss_get_ar(self._h, chain, _p(phi), C.byref(sig), _p(xtx), _p(xty),
                                          C.byref(yty), C.byref(n)))
        return dict(phi=phi, sigsq=sig.value, xtx=xtx, xty=xty, yty=yty.value, n=n.value)

    def ss_set_state_models(self, blocks):
        """a general list of state models, in the order they are added: dicts with kind
        (1 local level, 2 local linear trend, 3 seasonal, 4 autoregression, 5 static intercept,
        6 trig), nseasons, duration, t0, lags, df, sigma_guess, sigma_upper_limit, initial_sigma
        (one entry per variance parameter; none for kind 5), initial_phi, rotations (kind 6: the
        (cos, sin) pairs of the transition matrix), a0, P0 (tests/cases.py: general_spec)"""
        self._check(self.lib.ba_ss_clear_state_models(self._h))
        self._blocks = []
        for b in blocks:
            kind = int(b["kind"])
            ip = np.zeros(3, np.int32)
            if kind == 3:
                ip[:] = (b["nseasons"], b["duration"], b.get("t0", 0))
            elif kind == 4:
                ip[0] = b["lags"]
            elif kind == 6:
                ip[0] = len(b["rotations"]) // 2
            elif kind == 7:
                ip[0], ip[1] = int(b.get("force_stationary", 1)), int(b.get("force_positive", 0))
            arrs = [np.ascontiguousarray(b[k], dtype=np.float64) for k in
                    ("df", "sigma_guess", "sigma_upper_limit", "initial_sigma")]
            ph = np.ascontiguousarray(b["rotations"] if kind == 6 else
                                      (b["slope_priors"] if kind == 7 else b.get("initial_phi", np.zeros(0))),
                                      dtype=np.float64)
            a0 = np.ascontiguousarray(b["a0"], dtype=np.float64)
            p0 = np.ascontiguousarray(b["P0"], dtype=np.float64)
            self._check(self.lib.ba_ss_add_state_model(
                self._h, kind, ip.ctypes.data_as(C.POINTER(C.c_int32)),
                *[(_p(a) if a.size else None) for a in arrs],
                _p(ph) if (kind in (4, 6, 7) and ph.size) else None, _p(a0), _p(p0)))
            self._blocks.append(dict(kind=kind, nvar=2 if kind in (2, 7) else (0 if kind == 5 else 1),
                                     lags=int(ip[0]) if kind == 4 else 0))
        m, nb = C.c_int32(), C.c_int32()
        self._check(self.lib.ba_ss_state_dimension(self._h, C.byref(m), C.byref(nb)))
        self._ssm_dim = m.value
        self._ar_lags = 0

    def set_slot_limit(self, uniforms):
        """tests: a substream slot hands out `uniforms` numbers, then its spill stream (0: default)"""
        self._check(self.lib.ba_set_slot_limit(self._h, int(uniforms)))

    def ss_set_tuning(self, use_template_kernel=True, kernel=None):
        """kernel: 0 general, 1 the default choice,
        3 compiled for the shape (where it applies); 4 / 5: the local-level rounds as separate
        launches / as the persistent round kernel (the default)"""
        if kernel is None:
            kernel = 1 if use_template_kernel else 0
        self._check(self.lib.ba_ss_set_tuning(self._h, int(kernel)))

    def ss_get_state_model(self, chain, block, suf=True):
        """suf=False: the variance parameters / coefficients only"""
        b = self._blocks[block]
        nv, L = b["nvar"], b["lags"]
        var, n, ss = np.zeros(nv), np.zeros(nv), np.zeros(nv)
        out = dict(variances=var, suf_n=n, suf_ss=ss)
        if b["kind"] == 7:
            # a semilocal linear trend: (phi, mu) of the slope model; its Ar1Suf (sumsq, sum, cross,
            # n, first, last) with suf=True
            phi, a1 = np.zeros(2), np.zeros(6)
            self._check(self.lib.ba_ss_get_state_model(self._h, chain, block, _p(var), _p(n) if suf else None,
                                                       _p(ss) if suf else None, _p(phi), _p(a1) if suf else None,
                                                       None, None, None))
            out["phi"] = phi
            if suf:
                out["ar1_suf"] = a1
            return out
        if not suf:
            phi = np.zeros(L) if L else None
            self._check(self.lib.ba_ss_get_state_model(self._h, chain, block, _p(var), None, None,
                                                       _p(phi), None, None, None, None))
            if L:
                out["phi"] = phi
            return out
        if L:
            phi, xtx, xty = np.zeros(L), np.zeros((L, L)), np.zeros(L)
            yty, an = C.c_double(), C.c_double()
            self._check(self.lib.ba_ss_get_state_model(
                self._h, chain, block, _p(var), _p(n), _p(ss), _p(phi), _p(xtx), _p(xty),
                C.cast(C.byref(yty), _dp), C.cast(C.byref(an), _dp)))
            out.update(phi=phi, xtx=xtx, xty=xty, yty=yty.value, n=an.value)
        else:
            self._check(self.lib.ba_ss_get_state_model(self._h, chain, block, _p(var), _p(n), _p(ss),
                                                       None, None, None, None, None))
        return out

    def ss_get_state_draw(self, chain):
        st = np.zeros((self.T, self._ssm_dim))
        self._check(self.lib.ba_ss_get_state_draw(self._h, chain, _p(st)))
        return st

    def ss_get_structural(self, chain):
        st = np.zeros((self.T, self._ssm_dim))
        var, n, ss = np.zeros(3), np.zeros(3), np.zeros(3)
        self._check(self.lib.ba_ss_get_structural(self._h, chain, _p(st), _p(var), _p(n), _p(ss)))
        return dict(state=st, variances=var, suf_n=n, suf_ss=ss)

    def ss_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_ss_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def ss_set_lookahead(self, lookahead, chains=None):
        """ba_ss_draw_next enqueues `lookahead` rounds at a time; chains: whose state path
        is recorded (default: chain 0)"""
        if chains is not None:
            arr = np.ascontiguousarray(chains, dtype=np.int64)
            self._check(self.lib.ba_ss_lookahead_chains(self._h, len(arr),
                                                        arr.ctypes.data_as(C.POINTER(C.c_int64))))
        self._check(self.lib.ba_ss_set_lookahead(self._h, int(lookahead)))

    def ss_draw_next(self):
        self._check(self.lib.ba_ss_draw_next(self._h))

    def ss_impute_state(self):
        self._check(self.lib.ba_ss_impute_state(self._h))
        self.sync()

    def ss_forecast(self, newX):
        """one predictive draw of the next len(newX) observations per chain"""
        h = newX.shape[0]
        out = np.zeros((self.chains, h))
        self._check(self.lib.ba_ss_forecast(self._h, h, _p(_fcol(newX)), _p(out)))
        return out

    def ss_get_state(self, chain, state=True, suf=True):
        """state=False / suf=False: do not ask for the state path / the level model's
        sufficient statistics (NULL pointers)"""
        st = np.zeros(self.T) if state else None
        ls, n, ss = C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.ba_ss_get_state(self._h, chain, _p(st), C.byref(ls),
                                             C.byref(n) if suf else None, C.byref(ss) if suf else None))
        return dict(state=st, level_sigsq=ls.value, level_n=n.value, level_sumsq=ss.value)

    def ss_set_level_sigsq(self, sigsq, chain=-1):
        self._check(self.lib.ba_ss_set_level_sigsq(self._h, chain, sigsq))

    def ss_get_chain_suf(self, chain):
        xty = np.zeros(self.p)
        yty, n = C.c_double(), C.c_double()
        self._check(self.lib.ba_ss_get_chain_suf(self._h, chain, _p(xty), C.byref(yty),
                                                 C.byref(n)))
        return dict(xty=xty, yty=yty.value, n=n.value)


class Group:
    """Several engines behind one handle (ba_group_*): one per entry of `devices`, chains
    sharded by global id; the data build and the summaries are the two collectives."""

    def __init__(self, devices, chains_per_device, seed=8675309):
        self.lib = load_library()
        dv = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        self._h = None
        rc = self.lib.ba_group_create(dv, len(devices), chains_per_device, seed, C.byref(h))
        if rc != 0:
            raise BoomAmdError(rc, self.lib.ba_group_last_error().decode())
        self._h = h
        self.size = len(devices)
        self.chains_per_device = chains_per_device
        self.p = 0
        self.engines = [Engine(chains_per_device, _borrowed=self.lib.ba_group_engine(h, i))
                        for i in range(self.size)]

    def _check(self, rc):
        if rc != 0:
            raise BoomAmdError(rc, self.lib.ba_group_last_error().decode())

    def close(self):
        if self._h is not None:
            for e in self.engines:
                e.close()
            self.lib.ba_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, fn):
        """fn(engine) on every engine of the group through ba_group_call (the C-ABI's own
        loop: stops at the first error)"""
        by_handle = {e._h.value: e for e in self.engines}
        err = []

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
        def tramp(h, _):
            try:
                fn(by_handle[h])
                return 0
            except BoomAmdError as ex:
                err.append(ex)
                return ex.args[0] if ex.args and isinstance(ex.args[0], int) and ex.args[0] else -1
        rc = self.lib.ba_group_call(self._h, C.cast(tramp, C.c_void_p), None)
        if err:
            raise err[0]
        self._check(rc)

    def ss_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_group_ss_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def ss_draw_next(self):
        self._check(self.lib.ba_group_ss_draw_next(self._h))

    def logit_sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_group_logit_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def locate(self, global_chain):
        e, c = C.c_int32(), C.c_int64()
        self._check(self.lib.ba_group_locate(self._h, global_chain, C.byref(e), C.byref(c)))
        return e.value, c.value

    def build_suf_from_xy(self, X, y):
        n, p = X.shape
        self._check(self.lib.ba_group_build_suf_from_xy(self._h, n, p, _p(_fcol(X)), _p(_f64(y))))
        self.p = p
        for e in self.engines:
            e.p = p

    def set_priors(self, b, ominv, pi, prior_df, sigma_guess, max_model_size=-1,
                   sigma_upper_limit=float("inf")):
        self._check(self.lib.ba_group_set_priors(self._h, _p(_f64(b)), _p(_fcol(ominv)), _p(_f64(pi)),
                                                 int(max_model_size), prior_df, sigma_guess,
                                                 sigma_upper_limit))

    def set_state(self, gamma, beta=None, sigsq=1.0):
        g = np.ascontiguousarray(gamma, dtype=np.uint8)
        bb = None if beta is None else _f64(beta)
        self._check(self.lib.ba_group_set_state(self._h, _b(g), _p(bb), sigsq))

    def sweep(self, nsweeps=1, sync=True):
        self._check(self.lib.ba_group_sweep(self._h, nsweeps))
        if sync:
            self.sync()

    def sync(self):
        self._check(self.lib.ba_group_sync(self._h))

    def reset_summaries(self):
        self._check(self.lib.ba_group_reset_summaries(self._h))

    def get_summaries(self):
        p = self.p
        inc, bs, bs2 = np.zeros(p), np.zeros(p), np.zeros(p)
        sc = np.zeros(SUMMARY_SCALARS)
        blocks = np.zeros((self.size, 3 * p + SUMMARY_SCALARS))
        self._check(self.lib.ba_group_get_summaries(self._h, _p(inc), _p(bs), _p(bs2), _p(sc), _p(blocks)))
        return dict(inclusion_count=inc, beta_sum=bs, beta_sumsq=bs2, sweeps=sc[0], sigsq_sum=sc[1],
                    k_sum=sc[3], accepts=sc[4], proposals=sc[5], min_margin=sc[6], blocks=blocks)
