"""ctypes bindings for the two CHECKERS (test infrastructure only):

* ``Oracle``  -- oracle/libboomoracle.so, our clean-room C restatement
* ``Ref``     -- oracle/_ref/libboomref.so, the unmodified BOOM reference
                 (exists only where it was built: the build container, or the
                 GPU box via the prebuilt .so that travels with the snapshot)

Nothing under boom_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libboomoracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libboomref.so")

c_double_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)
c_int_p = C.POINTER(C.c_int)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _u8(a):
    return None if a is None else a.ctypes.data_as(c_u8_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def fcol(a):
    """column-major flat copy of a 2-d array"""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).ravel()


def build_oracle():
    src = os.path.join(ORACLE_DIR, "boom_oracle.c")
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"],
                              stdout=subprocess.DEVNULL)
    return ORACLE_SO


class BoRng(C.Structure):
    _fields_ = [("kind", C.c_int), ("mt", C.c_uint64 * 312), ("mti", C.c_int),
                ("seed", C.c_uint64), ("chain", C.c_uint32),
                ("stream", C.c_uint32), ("pos", C.c_uint64),
                ("slot_stride", C.c_uint64), ("slot", C.c_uint64),
                ("limit", C.c_uint64), ("spill", C.c_uint64)]


class RefSsvsOptions(C.Structure):
    _fields_ = [("max_model_size", C.c_int64), ("sigma_upper_limit", C.c_double),
                ("swap_threshold", C.c_double), ("max_flips", C.c_int),
                ("draw_beta", C.c_int), ("draw_sigma", C.c_int)]


class RefSsOptions(C.Structure):
    _fields_ = [("level_df", C.c_double), ("level_sigma_guess", C.c_double),
                ("level_sigma_upper_limit", C.c_double),
                ("initial_state_mean", C.c_double),
                ("initial_state_variance", C.c_double),
                ("initial_level_sigma", C.c_double)]


def ssvs_options(max_model_size=-1, sigma_upper_limit=float("inf"),
                 swap_threshold=0.8, max_flips=-1, draw_beta=1, draw_sigma=1):
    return dict(max_model_size=int(max_model_size),
                sigma_upper_limit=float(sigma_upper_limit),
                swap_threshold=float(swap_threshold), max_flips=int(max_flips),
                draw_beta=int(draw_beta), draw_sigma=int(draw_sigma))


# ---------------------------------------------------------------------------
class Oracle:
    def set_slot_limit(self, uniforms):
        """tests of the spill streams: a substream slot serves `uniforms` numbers (0: its stride)"""
        self.lib.bo_set_slot_limit(int(uniforms))

    def __init__(self):
        self.lib = L = C.CDLL(build_oracle())
        L.bo_set_slot_limit.argtypes = [C.c_int]
        L.bo_set_slot_limit.restype = None
        L.bo_unif.restype = C.c_double
        L.bo_norm_rand.restype = C.c_double
        L.bo_exp_rand.restype = C.c_double
        L.bo_rgamma.restype = C.c_double
        L.bo_rgamma.argtypes = [C.c_void_p, C.c_double, C.c_double, c_int_p]
        L.bo_rtrun_gamma.restype = C.c_double
        L.bo_rtrun_gamma.argtypes = [C.c_void_p, C.c_double, C.c_double,
                                     C.c_double, c_int_p]
        L.bo_seed_rng.restype = C.c_uint64
        L.bo_rng_seed_mt.argtypes = [C.c_void_p, C.c_uint64]
        L.bo_rng_seed_philox.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32,
                                         C.c_uint32, C.c_uint64]
        L.bo_spd_logdet.restype = C.c_double
        L.bo_spd_mdist.restype = C.c_double
        L.bo_ssvs_create.restype = C.c_void_p
        L.bo_ssvs_create.argtypes = [C.c_int, c_double_p, c_double_p, C.c_double,
                                     C.c_double, C.c_double, c_double_p,
                                     c_double_p, c_double_p, C.c_double,
                                     C.c_double, c_double_p]
        L.bo_ssvs_destroy.argtypes = [C.c_void_p]
        L.bo_ssvs_set_options.argtypes = [C.c_void_p, C.c_int64, C.c_double,
                                          C.c_double, C.c_int, C.c_int, C.c_int]
        L.bo_ssvs_set_state.argtypes = [C.c_void_p, c_u8_p, c_double_p,
                                        C.c_double]
        L.bo_ssvs_get_state.argtypes = [C.c_void_p, c_u8_p, c_double_p,
                                        c_double_p]
        L.bo_ssvs_get_perm.argtypes = [C.c_void_p, c_int_p]
        L.bo_ssvs_rng.restype = C.c_void_p
        L.bo_ssvs_rng.argtypes = [C.c_void_p]
        L.bo_ssvs_draw.argtypes = [C.c_void_p]
        L.bo_ssvs_min_margin.restype = C.c_double
        L.bo_ssvs_min_margin.argtypes = [C.c_void_p]
        L.bo_ssvs_log_model_prob.restype = C.c_double
        L.bo_ssvs_log_model_prob.argtypes = [C.c_void_p, c_u8_p, c_int_p]
        L.bo_ssvs_run_chains.argtypes = [
            C.c_int, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double,
            c_double_p, c_double_p, c_double_p, C.c_double, C.c_double,
            c_double_p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_uint64,
            C.c_int, C.c_int, C.c_int, c_u8_p, c_double_p, c_double_p]
        L.bo_ss_create.restype = C.c_void_p
        L.bo_ss_create.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p,
                                   c_u8_p, c_double_p, c_double_p, C.c_double,
                                   C.c_double, c_double_p, C.c_double,
                                   C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double]
        L.bo_ss_destroy.argtypes = [C.c_void_p]
        L.bo_ss_regression.restype = C.c_void_p
        L.bo_ss_regression.argtypes = [C.c_void_p]
        L.bo_ss_level_rng.restype = C.c_void_p
        L.bo_ss_level_rng.argtypes = [C.c_void_p]
        L.bo_ss_state_rng.restype = C.c_void_p
        L.bo_ss_state_rng.argtypes = [C.c_void_p]
        L.bo_ss_set_level_sigsq.argtypes = [C.c_void_p, C.c_double]
        L.bo_ss_level_sigsq.restype = C.c_double
        L.bo_ss_level_sigsq.argtypes = [C.c_void_p]
        L.bo_ss_state.restype = c_double_p
        L.bo_ss_state.argtypes = [C.c_void_p]
        L.bo_ss_level_suf.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.bo_ss_impute_state.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_ss_draw.argtypes = [C.c_void_p]
        L.bo_probit_create.restype = C.c_void_p
        L.bo_probit_create.argtypes = [C.c_int, C.c_int] + [c_double_p] * 6 + [C.c_int]
        L.bo_probit_destroy.argtypes = [C.c_void_p]
        L.bo_probit_sss.restype = C.c_void_p
        L.bo_probit_sss.argtypes = [C.c_void_p]
        L.bo_probit_imputer_rng.restype = C.c_void_p
        L.bo_probit_imputer_rng.argtypes = [C.c_void_p]
        L.bo_probit_use_substreams.argtypes = [C.c_void_p, C.c_int]
        L.bo_probit_draw.argtypes = [C.c_void_p]
        L.bo_rtrun_norm.restype = C.c_double
        L.bo_rtrun_norm.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int,
                                    C.c_void_p]
        L.bo_logit_create.restype = C.c_void_p
        L.bo_logit_create.argtypes = [C.c_int, C.c_int] + [c_double_p] * 6 + [C.c_int]
        L.bo_logit_destroy.argtypes = [C.c_void_p]
        L.bo_logit_sss.restype = C.c_void_p
        L.bo_logit_sss.argtypes = [C.c_void_p]
        L.bo_logit_worker_rng.restype = C.c_void_p
        L.bo_logit_worker_rng.argtypes = [C.c_void_p]
        L.bo_logit_use_substreams.argtypes = [C.c_void_p, C.c_int]
        L.bo_logit_get_suf.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.bo_logit_draw.argtypes = [C.c_void_p]
        L.bo_ssm_create.restype = C.c_void_p
        L.bo_ssm_create.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p,
                                    C.POINTER(C.c_uint8), c_double_p, c_double_p,
                                    C.c_double, C.c_double, c_double_p, C.c_int,
                                    C.c_int, c_double_p, c_double_p, c_double_p,
                                    c_double_p, c_double_p, c_double_p]
        L.bo_ssm_destroy.argtypes = [C.c_void_p]
        L.bo_ssm_regression.restype = C.c_void_p
        L.bo_ssm_regression.argtypes = [C.c_void_p]
        L.bo_ssm_variance_rng.restype = C.c_void_p
        L.bo_ssm_variance_rng.argtypes = [C.c_void_p, C.c_int]
        L.bo_ssm_state_rng.restype = C.c_void_p
        L.bo_ssm_state_rng.argtypes = [C.c_void_p]
        L.bo_ssm_state_dimension.argtypes = [C.c_void_p]
        L.bo_ssm_state.restype = c_double_p
        L.bo_ssm_state.argtypes = [C.c_void_p]
        L.bo_ssm_get_variances.argtypes = [C.c_void_p, c_double_p]
        L.bo_ssm_set_variances.argtypes = [C.c_void_p, c_double_p]
        L.bo_ssm_get_suf.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.bo_ssm_impute_state.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_ssm_draw.argtypes = [C.c_void_p]
        L.bo_ssm_add_ar.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                    C.c_double, c_double_p, c_double_p, c_double_p]
        L.bo_ssm_ar_rng.restype = C.c_void_p
        L.bo_ssm_ar_rng.argtypes = [C.c_void_p]
        L.bo_ssm_get_ar.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.bo_ssm_get_ar_suf.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p,
                                        c_double_p]
        L.bo_test_ar_check_stationary.argtypes = [C.c_int, c_double_p]

    # -- RNG -----------------------------------------------------------------
    def rng_mt(self, seed):
        r = BoRng()
        self.lib.bo_rng_seed_mt(C.byref(r), int(seed))
        return r

    def rng_philox(self, seed, chain=0, stream=0, pos=0):
        r = BoRng()
        self.lib.bo_rng_seed_philox(C.byref(r), int(seed), chain, stream, pos)
        return r

    def uniforms(self, rng, n):
        return np.array([self.lib.bo_unif(C.byref(rng)) for _ in range(n)])

    def norms(self, rng, n):
        return np.array([self.lib.bo_norm_rand(C.byref(rng)) for _ in range(n)])

    def exps(self, rng, n):
        return np.array([self.lib.bo_exp_rand(C.byref(rng)) for _ in range(n)])

    def gammas(self, rng, a, b, n):
        st = C.c_int(0)
        out = np.array([self.lib.bo_rgamma(C.byref(rng), a, b, C.byref(st))
                        for _ in range(n)])
        assert st.value == 0
        return out

    def trun_gammas(self, rng, a, b, cut, n):
        st = C.c_int(0)
        out = np.array([self.lib.bo_rtrun_gamma(C.byref(rng), a, b, cut,
                                                C.byref(st)) for _ in range(n)])
        assert st.value == 0
        return out

    def seed_rngs(self, rng, n):
        return np.array([self.lib.bo_seed_rng(C.byref(rng)) for _ in range(n)],
                        dtype=np.uint64)

    def random_ints(self, rng, lo, hi, n):
        return np.array([self.lib.bo_random_int(C.byref(rng), lo, hi)
                         for _ in range(n)], dtype=np.int32)

    def shuffles(self, rng, p, nrep):
        v = np.arange(p, dtype=np.int32)
        out = np.zeros((nrep, p), dtype=np.int32)
        for r in range(nrep):
            self.lib.bo_shuffle(C.byref(rng), _ip(v), p)
            out[r] = v
        return out

    def rmultis(self, rng, prob, n):
        prob = f64(prob)
        st = C.c_int(0)
        return np.array([self.lib.bo_rmulti(C.byref(rng), _dp(prob), len(prob),
                                            C.byref(st)) for _ in range(n)],
                        dtype=np.int32)

    # -- LinAlg ----------------------------------------------------------------
    def chol(self, A):
        n = A.shape[0]
        a = fcol(A)
        L = np.zeros(n * n)
        ok = self.lib.bo_chol(n, _dp(a), _dp(L))
        return L.reshape(n, n).T.copy(), bool(ok)

    def logdet(self, A):
        n = A.shape[0]
        a = fcol(A)
        ok = C.c_int(0)
        v = self.lib.bo_spd_logdet(n, _dp(a), C.byref(ok))
        return v, bool(ok.value)

    def solve(self, A, rhs):
        n = A.shape[0]
        a = fcol(A)
        rhs = f64(rhs)
        x = np.zeros(n)
        ok = self.lib.bo_spd_solve(n, _dp(a), _dp(rhs), _dp(x))
        return x, bool(ok)

    def mdist(self, A, x):
        a = fcol(A)
        x = f64(x)
        return self.lib.bo_spd_mdist(A.shape[0], _dp(a), _dp(x))

    def neregsuf(self, X, y):
        n, p = X.shape
        xc = fcol(X)
        y = f64(y)
        xtx = np.zeros(p * p)
        xty = np.zeros(p)
        yty = C.c_double()
        sumy = C.c_double()
        xsum = np.zeros(p)
        self.lib.bo_neregsuf(n, p, _dp(xc), _dp(y), _dp(xtx), _dp(xty),
                             C.byref(yty), C.byref(sumy), _dp(xsum))
        return dict(xtx=xtx.reshape(p, p).T.copy(), xty=xty, yty=yty.value,
                    n=float(n), sumy=sumy.value, xsum=xsum)

    # -- SSVS --------------------------------------------------------------------
    def ssvs_create(self, suf, prior):
        p = len(suf["xty"])
        h = self.lib.bo_ssvs_create(
            p, _dp(fcol(suf["xtx"])), _dp(f64(suf["xty"])), suf["yty"],
            suf["n"], suf["sumy"], _dp(f64(suf["xsum"])), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), prior["df"], prior["sigma_guess"],
            _dp(f64(prior["pi"])))
        return h

    def ssvs_run(self, suf, prior, opts, rng_setup, init_gamma, nsweeps,
                 init_beta=None, init_sigsq=1.0, want_margin=False):
        """rng_setup: ('mt', global_seed) -> sampler seeded by seed_rng(global)
        as the reference wrappers do; or ('philox', seed, chain)."""
        p = len(suf["xty"])
        h = self.ssvs_create(suf, prior)
        self.lib.bo_ssvs_set_options(h, opts["max_model_size"],
                                     opts["sigma_upper_limit"],
                                     opts["swap_threshold"], opts["max_flips"],
                                     opts["draw_beta"], opts["draw_sigma"])
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        b0 = f64(init_beta) if init_beta is not None else np.zeros(p)
        self.lib.bo_ssvs_set_state(h, _u8(g0), _dp(b0), float(init_sigsq))
        rp = self.lib.bo_ssvs_rng(h)
        if rng_setup[0] == "mt":
            glob = self.rng_mt(rng_setup[1])
            seed = self.lib.bo_seed_rng(C.byref(glob))
            self.lib.bo_rng_seed_mt(rp, seed)
        else:
            self.lib.bo_rng_seed_philox(rp, int(rng_setup[1]), int(rng_setup[2]),
                                        0, 0)
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        s = C.c_double()
        status = 0
        for i in range(nsweeps):
            status = self.lib.bo_ssvs_draw(h)
            if status:
                break
            self.lib.bo_ssvs_get_state(h, _u8(g), _dp(b), C.byref(s))
            gam[i] = g
            beta[i] = b
            sig[i] = s.value
        margin = self.lib.bo_ssvs_min_margin(h)
        self.lib.bo_ssvs_destroy(h)
        out = dict(gamma=gam, beta=beta, sigsq=sig, status=status)
        if want_margin:
            out["min_margin"] = margin
        return out

    def ssvs_run_priors_changed(self, suf, prior, prior2, change_at, opts, rng_setup, init_gamma, nsweeps):
        """ssvs_run, with the priors replaced by prior2 before draw `change_at` (ctor #5's
        prior objects modified under the sampler, BregVsSampler.hpp:98-101)"""
        p = len(suf["xty"])
        h = self.ssvs_create(suf, prior)
        self.lib.bo_ssvs_set_options(h, opts["max_model_size"], opts["sigma_upper_limit"],
                                     opts["swap_threshold"], opts["max_flips"],
                                     opts["draw_beta"], opts["draw_sigma"])
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self.lib.bo_ssvs_set_state(h, _u8(g0), _dp(np.zeros(p)), 1.0)
        self.lib.bo_rng_seed_philox(self.lib.bo_ssvs_rng(h), int(rng_setup[1]), int(rng_setup[2]), 0, 0)
        self.lib.bo_ssvs_set_priors.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_double,
                                                C.c_double, c_double_p]
        self.lib.bo_ssvs_set_priors.restype = None
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        g, b, s = np.zeros(p, dtype=np.uint8), np.zeros(p), C.c_double()
        status = 0
        for i in range(nsweeps):
            if i == change_at:
                self.lib.bo_ssvs_set_priors(h, _dp(f64(prior2["b"])), _dp(fcol(prior2["ominv"])),
                                            float(prior2["df"]), float(prior2["sigma_guess"]),
                                            _dp(f64(prior2["pi"])))
            status = self.lib.bo_ssvs_draw(h)
            if status:
                break
            self.lib.bo_ssvs_get_state(h, _u8(g), _dp(b), C.byref(s))
            gam[i], beta[i], sig[i] = g, b, s.value
        margin = self.lib.bo_ssvs_min_margin(h)
        self.lib.bo_ssvs_destroy(h)
        return dict(gamma=gam, beta=beta, sigsq=sig, status=status, min_margin=margin)

    def log_model_prob(self, suf, prior, gammas, max_model_size=-1):
        h = self.ssvs_create(suf, prior)
        self.lib.bo_ssvs_set_options(h, max_model_size, float("inf"), 0.8, -1,
                                     1, 1)
        out = []
        for g in gammas:
            g = np.ascontiguousarray(g, dtype=np.uint8)
            st = C.c_int(0)
            out.append(self.lib.bo_ssvs_log_model_prob(h, _u8(g), C.byref(st)))
        self.lib.bo_ssvs_destroy(h)
        return np.array(out)

    def logpri(self, suf, prior, gammas, betas, sigsqs, max_model_size=-1):
        """BregVsSampler::logpri() at the given states"""
        h = self.ssvs_create(suf, prior)
        self.lib.bo_ssvs_set_options(h, max_model_size, float("inf"), 0.8, -1,
                                     1, 1)
        self.lib.bo_ssvs_logpri.restype = C.c_double
        self.lib.bo_ssvs_logpri.argtypes = [C.c_void_p]
        out = []
        for g, b, s2 in zip(gammas, betas, sigsqs):
            g = np.ascontiguousarray(g, dtype=np.uint8)
            self.lib.bo_ssvs_set_state(h, _u8(g), _dp(f64(b)), C.c_double(float(s2)))
            out.append(self.lib.bo_ssvs_logpri(h))
        self.lib.bo_ssvs_destroy(h)
        return np.array(out)

    def prior_ctor1(self, suf, prior_nobs, expected_rsq, expected_model_size,
                    first_term_is_intercept):
        p = len(suf["xty"])
        b = np.zeros(p)
        om = np.zeros(p * p)
        pi = np.zeros(p)
        df = C.c_double()
        sg = C.c_double()
        self.lib.bo_breg_prior_ctor1.argtypes = [
            C.c_int, c_double_p, C.c_double, C.c_double, C.c_double, C.c_double,
            C.c_double, C.c_double, C.c_int, c_double_p, c_double_p, c_double_p,
            c_double_p, c_double_p]
        self.lib.bo_breg_prior_ctor1(p, _dp(fcol(suf["xtx"])), suf["yty"],
                                     suf["n"], suf["sumy"], prior_nobs,
                                     expected_rsq, expected_model_size,
                                     int(first_term_is_intercept), _dp(b),
                                     _dp(om), _dp(pi), C.byref(df), C.byref(sg))
        return dict(b=b, ominv=om.reshape(p, p).T.copy(), pi=pi, df=df.value,
                    sigma_guess=sg.value)

    def prior_ctor2(self, suf, prior_sigma_nobs, prior_sigma_guess,
                    prior_beta_nobs, diagonal_shrinkage,
                    prior_inclusion_probability, force_intercept):
        p = len(suf["xty"])
        b = np.zeros(p)
        om = np.zeros(p * p)
        pi = np.zeros(p)
        df = C.c_double()
        sg = C.c_double()
        self.lib.bo_breg_prior_ctor2.argtypes = [
            C.c_int, c_double_p, C.c_double, C.c_double, C.c_double, C.c_double,
            C.c_double, C.c_double, C.c_double, C.c_int, c_double_p, c_double_p,
            c_double_p, c_double_p, c_double_p]
        self.lib.bo_breg_prior_ctor2(p, _dp(fcol(suf["xtx"])), suf["n"],
                                     suf["sumy"], prior_sigma_nobs,
                                     prior_sigma_guess, prior_beta_nobs,
                                     diagonal_shrinkage,
                                     prior_inclusion_probability,
                                     int(force_intercept), _dp(b), _dp(om),
                                     _dp(pi), C.byref(df), C.byref(sg))
        return dict(b=b, ominv=om.reshape(p, p).T.copy(), pi=pi, df=df.value,
                    sigma_guess=sg.value)

    def adaptive_run(self, suf, prior, opts, rng_setup, init_gamma, nsweeps,
                     max_flips=-1, step_size=-1.0, target=-1.0, want_margin=False):
        """AdaptiveSpikeSlabRegressionSampler::draw() x nsweeps.  rng_setup:
        ('mt', global_seed) -> sampler seeded by seed_rng(global) like every
        PosteriorSampler; or ('philox', seed, chain) (stream 4)."""
        L = self.lib
        L.bo_adaptive_create.restype = C.c_void_p
        L.bo_adaptive_create.argtypes = [C.c_void_p]
        L.bo_adaptive_destroy.argtypes = [C.c_void_p]
        L.bo_adaptive_set_options.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.bo_adaptive_get_rates.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.bo_adaptive_min_margin.restype = C.c_double
        L.bo_adaptive_min_margin.argtypes = [C.c_void_p]
        L.bo_adaptive_min_multi_margin.restype = C.c_double
        L.bo_adaptive_min_multi_margin.argtypes = [C.c_void_p]
        L.bo_adaptive_draw.argtypes = [C.c_void_p]
        p = len(suf["xty"])
        h = self.ssvs_create(suf, prior)
        L.bo_ssvs_set_options(h, opts["max_model_size"], opts["sigma_upper_limit"],
                              opts["swap_threshold"], opts["max_flips"],
                              opts["draw_beta"], opts["draw_sigma"])
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        L.bo_ssvs_set_state(h, _u8(g0), _dp(np.zeros(p)), 1.0)
        rp = L.bo_ssvs_rng(h)
        if rng_setup[0] == "mt":
            glob = self.rng_mt(rng_setup[1])
            L.bo_rng_seed_mt(rp, L.bo_seed_rng(C.byref(glob)))
        else:
            L.bo_rng_seed_philox(rp, int(rng_setup[1]), int(rng_setup[2]), 4, 0)
        a = L.bo_adaptive_create(h)
        L.bo_adaptive_set_options(a, int(max_flips), float(step_size), float(target))
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        sv = C.c_double()
        status = 0
        for i in range(nsweeps):
            status = L.bo_adaptive_draw(a)
            if status:
                break
            L.bo_ssvs_get_state(h, _u8(g), _dp(b), C.byref(sv))
            gam[i] = g
            beta[i] = b
            sig[i] = sv.value
        birth = np.zeros(p)
        death = np.zeros(p)
        L.bo_adaptive_get_rates(a, _dp(birth), _dp(death))
        out = dict(gamma=gam, beta=beta, sigsq=sig, status=status, birth=birth,
                   death=death)
        if want_margin:
            out["min_margin"] = L.bo_adaptive_min_margin(a)
            out["min_multi_margin"] = L.bo_adaptive_min_multi_margin(a)
        L.bo_adaptive_destroy(a)
        L.bo_ssvs_destroy(h)
        return out

    def run_chains(self, suf, prior, opts, seed, chains, nsweeps, nthreads,
                   init_gamma, init_beta=None, init_sigsq=None):
        p = len(suf["xty"])
        gam = np.ascontiguousarray(
            np.broadcast_to(np.asarray(init_gamma, dtype=np.uint8),
                            (chains, p))).copy()
        beta = (np.zeros((chains, p)) if init_beta is None
                else f64(np.broadcast_to(init_beta, (chains, p))).copy())
        sig = (np.ones(chains) if init_sigsq is None
               else f64(np.broadcast_to(init_sigsq, (chains,))).copy())
        st = self.lib.bo_ssvs_run_chains(
            p, _dp(fcol(suf["xtx"])), _dp(f64(suf["xty"])), suf["yty"],
            suf["n"], suf["sumy"], _dp(f64(suf["xsum"])), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), prior["df"], prior["sigma_guess"],
            _dp(f64(prior["pi"])), opts["max_model_size"],
            opts["sigma_upper_limit"], opts["swap_threshold"],
            opts["max_flips"], int(seed), chains, nsweeps, nthreads, _u8(gam),
            _dp(beta), _dp(sig))
        return dict(gamma=gam, beta=beta, sigsq=sig, status=st)

    # -- SpikeSlabSampler (sigma^2 given) ---------------------------------------
    def _declare_sss(self):
        L = self.lib
        L.bo_sss_create.restype = C.c_void_p
        L.bo_sss_create.argtypes = [C.c_int, c_double_p, c_double_p, C.c_int,
                                    c_double_p, c_double_p, c_double_p]
        L.bo_sss_destroy.argtypes = [C.c_void_p]
        L.bo_sss_set_options.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.bo_sss_set_state.argtypes = [C.c_void_p, c_u8_p, c_double_p]
        L.bo_sss_get_state.argtypes = [C.c_void_p, c_u8_p, c_double_p]
        L.bo_sss_rng.restype = C.c_void_p
        L.bo_sss_rng.argtypes = [C.c_void_p]
        L.bo_sss_draw_model_indicators.argtypes = [C.c_void_p, C.c_double]
        L.bo_sss_draw_beta.argtypes = [C.c_void_p, C.c_double]

    def sss_run(self, xtx, xty, slab_kind, mu, prec, pi, rng_setup, init_gamma,
                sigsq_seq, max_model_size=-1, max_flips=-1):
        self._declare_sss()
        L = self.lib
        p = len(xty)
        h = L.bo_sss_create(p, _dp(fcol(xtx)), _dp(f64(xty)), int(slab_kind),
                            _dp(f64(mu)), _dp(fcol(prec)), _dp(f64(pi)))
        L.bo_sss_set_options(h, int(max_model_size), int(max_flips))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        L.bo_sss_set_state(h, _u8(g0), _dp(np.zeros(p)))
        rp = L.bo_sss_rng(h)
        if rng_setup[0] == "mt":
            L.bo_rng_seed_mt(rp, int(rng_setup[1]))
        else:
            L.bo_rng_seed_philox(rp, int(rng_setup[1]), int(rng_setup[2]), 3, 0)
        n = len(sigsq_seq)
        gam = np.zeros((n, p), dtype=np.uint8)
        beta = np.zeros((n, p))
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        status = 0
        for i, s2 in enumerate(sigsq_seq):
            status = L.bo_sss_draw_model_indicators(h, float(s2))
            if status:
                break
            status = L.bo_sss_draw_beta(h, float(s2))
            if status:
                break
            L.bo_sss_get_state(h, _u8(g), _dp(b))
            gam[i] = g
            beta[i] = b
        L.bo_sss_destroy(h)
        return dict(gamma=gam, beta=beta, status=status)

    # -- state space -----------------------------------------------------------
    def ss_create(self, y, X, observed, prior, ss):
        T, p = X.shape
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        self._keep = (obs,)
        return self.lib.bo_ss_create(
            T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), prior["df"], prior["sigma_guess"],
            _dp(f64(prior["pi"])), ss["level_df"], ss["level_sigma_guess"],
            ss["level_sigma_upper_limit"], ss["initial_state_mean"],
            ss["initial_state_variance"], ss["initial_level_sigma"])

    def ss_run(self, y, X, observed, prior, opts, ss, rng_setup, init_gamma,
               nsweeps, keep_state=None, prior2=None, change_at=-1):
        """prior2 / change_at: the regression's priors replaced before draw `change_at` (prior
        objects modified under the sampler)"""
        T, p = X.shape
        m = self.ss_create(y, X, observed, prior, ss)
        reg = self.lib.bo_ss_regression(m)
        self.lib.bo_ssvs_set_options(reg, opts["max_model_size"],
                                     opts["sigma_upper_limit"],
                                     opts["swap_threshold"], opts["max_flips"],
                                     opts["draw_beta"], opts["draw_sigma"])
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self.lib.bo_ssvs_set_state(reg, _u8(g0), _dp(np.zeros(p)), 1.0)
        rngs = [self.lib.bo_ssvs_rng(reg), self.lib.bo_ss_level_rng(m),
                self.lib.bo_ss_state_rng(m)]
        if rng_setup[0] == "mt":
            glob = self.rng_mt(rng_setup[1])
            for rp in rngs:  # construction order: regression, level, state
                self.lib.bo_rng_seed_mt(rp, self.lib.bo_seed_rng(C.byref(glob)))
        else:
            for sid, rp in enumerate(rngs):
                self.lib.bo_rng_seed_philox(rp, int(rng_setup[1]),
                                            int(rng_setup[2]), sid, 0)
            if rng_setup[0] == "philox_seq":
                # the state stream read in sequence, as the reference reads its RNG,
                # instead of one substream per normal (tests of the substream bridge)
                C.cast(rngs[2], C.POINTER(BoRng)).contents.slot_stride = 0
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        lev = np.zeros(nsweeps)
        state = np.zeros((nsweeps, T)) if keep_state is None else np.zeros((nsweeps, len(keep_state)))
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        s = C.c_double()
        status = 0
        for i in range(nsweeps):
            if prior2 is not None and i == change_at:
                self.lib.bo_ssvs_set_priors.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_double,
                                                        C.c_double, c_double_p]
                self.lib.bo_ssvs_set_priors.restype = None
                self.lib.bo_ssvs_set_priors(reg, _dp(f64(prior2["b"])), _dp(fcol(prior2["ominv"])),
                                            float(prior2["df"]), float(prior2["sigma_guess"]),
                                            _dp(f64(prior2["pi"])))
            status = self.lib.bo_ss_draw(m)
            if status:
                break
            self.lib.bo_ssvs_get_state(reg, _u8(g), _dp(b), C.byref(s))
            gam[i] = g
            beta[i] = b
            sig[i] = s.value
            lev[i] = self.lib.bo_ss_level_sigsq(m)
            st = np.ctypeslib.as_array(self.lib.bo_ss_state(m), (T,))
            state[i] = st if keep_state is None else st[keep_state]
        self.lib.bo_ss_destroy(m)
        return dict(gamma=gam, beta=beta, sigsq=sig, level_sigsq=lev,
                    state=state, status=status)

    def trun_norms(self, rng, mu, sigma, cut, above, n):
        st = C.c_int(0)
        out = np.array([self.lib.bo_rtrun_norm(C.byref(rng), mu, sigma, cut, int(above),
                                               C.byref(st)) for _ in range(n)])
        assert st.value == 0
        return out

    def probit_run(self, X, y, ntrials, slab, pi, rng_setup, init_gamma, init_beta, nsweeps,
                   clt_threshold=5, max_model_size=-1, max_flips=-1):
        """BinomialProbitSpikeSlabSampler (f3): slab = dict(mu, prec)"""
        n, p = X.shape
        self._declare_sss()
        m = self.lib.bo_probit_create(n, p, _dp(fcol(X)), _dp(f64(y)), _dp(f64(ntrials)),
                                      _dp(f64(slab["mu"])), _dp(fcol(slab["prec"])),
                                      _dp(f64(pi)), int(clt_threshold))
        sss = self.lib.bo_probit_sss(m)
        self.lib.bo_sss_set_options(sss, int(max_model_size), int(max_flips))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self.lib.bo_sss_set_state(sss, _u8(g0), _dp(f64(init_beta) * g0))
        if rng_setup[0] == "mt":
            # PosteriorSampler's rng is seeded from the global one
            glob = self.rng_mt(rng_setup[1])
            self.lib.bo_rng_seed_mt(self.lib.bo_sss_rng(sss),
                                    self.lib.bo_seed_rng(C.byref(glob)))
        else:
            seed, chain = int(rng_setup[1]), int(rng_setup[2])
            self.lib.bo_rng_seed_philox(self.lib.bo_sss_rng(sss), seed, chain, 3, 0)
            self.lib.bo_rng_seed_philox(self.lib.bo_probit_imputer_rng(m), seed, chain, 8, 0)
            self.lib.bo_probit_use_substreams(m, 1)
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        status = 0
        for i in range(nsweeps):
            status = self.lib.bo_probit_draw(m)
            if status:
                break
            self.lib.bo_sss_get_state(sss, _u8(g), _dp(b))
            gam[i] = g
            beta[i] = b
        self.lib.bo_probit_destroy(m)
        return dict(gamma=gam, beta=beta, status=status)

    def logit_run(self, X, y, ntrials, slab, pi, rng_setup, init_gamma, init_beta, nsweeps,
                  clt_threshold=5, max_model_size=-1, max_flips=-1, want_suf=False, imputer=0):
        """BinomialLogitSpikeSlabSampler (f3): slab = dict(mu, prec)"""
        n, p = X.shape
        self._declare_sss()
        m = self.lib.bo_logit_create(n, p, _dp(fcol(X)), _dp(f64(y)), _dp(f64(ntrials)),
                                     _dp(f64(slab["mu"])), _dp(fcol(slab["prec"])),
                                     _dp(f64(pi)), int(clt_threshold))
        sss = self.lib.bo_logit_sss(m)
        self.lib.bo_sss_set_options(sss, int(max_model_size), int(max_flips))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self.lib.bo_sss_set_state(sss, _u8(g0), _dp(f64(init_beta) * g0))
        if rng_setup[0] == "mt":
            # the sampler's rng is seeded from the global one; its imputation worker's
            # from the sampler's (at construction)
            glob = self.rng_mt(rng_setup[1])
            srng = self.lib.bo_sss_rng(sss)
            self.lib.bo_rng_seed_mt(srng, self.lib.bo_seed_rng(C.byref(glob)))
            self.lib.bo_rng_seed_mt(self.lib.bo_logit_worker_rng(m),
                                    self.lib.bo_seed_rng(C.c_void_p(srng)))
        else:
            seed, chain = int(rng_setup[1]), int(rng_setup[2])
            self.lib.bo_rng_seed_philox(self.lib.bo_sss_rng(sss), seed, chain, 3, 0)
            self.lib.bo_rng_seed_philox(self.lib.bo_logit_worker_rng(m), seed, chain,
                                        10 if imputer else 9, 0)
            self.lib.bo_logit_use_substreams(m, 1)
        if imputer:
            self.lib.bo_logit_set_imputer.argtypes = [C.c_void_p, C.c_int]
            self.lib.bo_logit_set_imputer(m, int(imputer))
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        xtx = np.zeros((nsweeps, p, p)) if want_suf else None
        xty = np.zeros((nsweeps, p)) if want_suf else None
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        status = 0
        for i in range(nsweeps):
            status = self.lib.bo_logit_draw(m)
            if status:
                break
            self.lib.bo_sss_get_state(sss, _u8(g), _dp(b))
            gam[i] = g
            beta[i] = b
            if want_suf:
                self.lib.bo_logit_get_suf(m, _dp(xtx[i]), _dp(xty[i]))
        self.lib.bo_logit_destroy(m)
        return dict(gamma=gam, beta=beta, status=status, xtx=xtx, xty=xty)

    def poisson_run(self, X, y, exposure, slab, pi, mix, rng_setup, init_gamma, init_beta, nsweeps,
                    max_model_size=-1, max_flips=-1, want_suf=False):
        """PoissonRegressionSpikeSlabSampler (f3): slab = dict(mu, prec); mix = the reference
        table's mixtures for the counts in the data (poisson_mixtures / the golden fixture)"""
        n, p = X.shape
        self._declare_sss()
        L = self.lib
        L.bo_poisson_create.restype = C.c_void_p
        L.bo_poisson_create.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                                        c_double_p, c_double_p, C.c_int, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int), c_double_p, c_double_p, c_double_p, C.c_int64]
        for name in ("bo_poisson_sss", "bo_poisson_worker_rng"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.bo_poisson_destroy.argtypes = [C.c_void_p]
        L.bo_poisson_use_substreams.argtypes = [C.c_void_p, C.c_int]
        L.bo_poisson_draw.argtypes = [C.c_void_p]
        L.bo_poisson_get_suf.argtypes = [C.c_void_p, c_double_p, c_double_p]
        counts = np.ascontiguousarray(mix["counts"], dtype=np.int64)
        ncomp = np.ascontiguousarray(mix["ncomp"], dtype=np.int32)
        m = L.bo_poisson_create(n, p, _dp(fcol(X)), _dp(f64(y)), _dp(f64(exposure)), _dp(f64(slab["mu"])),
                                _dp(fcol(slab["prec"])), _dp(f64(pi)), len(counts),
                                counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                ncomp.ctypes.data_as(C.POINTER(C.c_int)), _dp(f64(mix["mu"])),
                                _dp(f64(mix["sigma"])), _dp(f64(mix["weight"])), int(mix["largest_index"]))
        sss = L.bo_poisson_sss(m)
        L.bo_sss_set_options(sss, int(max_model_size), int(max_flips))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        L.bo_sss_set_state(sss, _u8(g0), _dp(f64(init_beta) * g0))
        if rng_setup[0] == "mt":
            glob = self.rng_mt(rng_setup[1])
            srng = L.bo_sss_rng(sss)
            L.bo_rng_seed_mt(srng, L.bo_seed_rng(C.byref(glob)))
            L.bo_rng_seed_mt(L.bo_poisson_worker_rng(m), L.bo_seed_rng(C.c_void_p(srng)))
        else:
            seed, chain = int(rng_setup[1]), int(rng_setup[2])
            L.bo_rng_seed_philox(L.bo_sss_rng(sss), seed, chain, 3, 0)
            L.bo_rng_seed_philox(L.bo_poisson_worker_rng(m), seed, chain, 11, 0)
            L.bo_poisson_use_substreams(m, 1)
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        xtx = np.zeros((nsweeps, p, p)) if want_suf else None
        xty = np.zeros((nsweeps, p)) if want_suf else None
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        status = 0
        for i in range(nsweeps):
            status = L.bo_poisson_draw(m)
            if status:
                break
            L.bo_sss_get_state(sss, _u8(g), _dp(b))
            gam[i] = g
            beta[i] = b
            if want_suf:
                L.bo_poisson_get_suf(m, _dp(xtx[i]), _dp(xty[i]))
        L.bo_poisson_destroy(m)
        return dict(gamma=gam, beta=beta, status=status, xtx=xtx, xty=xty)

    def ssm_forecast(self, rng, newX, beta, sigsq_obs, trend, nseasons, sigsq, final_state,
                     ar_phi=None, ar_sigsq=0.0):
        h, p = newX.shape
        out = np.zeros(h)
        if ar_phi is not None:
            self.lib.bo_ssm_simulate_forecast_ar.argtypes = [
                C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, C.c_double, C.c_int,
                C.c_int, c_double_p, C.c_int, c_double_p, C.c_double, c_double_p, c_double_p]
            self.lib.bo_ssm_simulate_forecast_ar(
                C.byref(rng), h, p, _dp(fcol(newX)), _dp(f64(beta)), float(sigsq_obs),
                int(trend), int(nseasons), _dp(f64(sigsq)), len(ar_phi), _dp(f64(ar_phi)),
                float(ar_sigsq), _dp(f64(final_state)), _dp(out))
            return out
        self.lib.bo_ssm_simulate_forecast.argtypes = [
            C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, C.c_double, C.c_int, C.c_int,
            c_double_p, c_double_p, c_double_p]
        self.lib.bo_ssm_simulate_forecast(C.byref(rng), h, p, _dp(fcol(newX)), _dp(f64(beta)),
                                          float(sigsq_obs), int(trend), int(nseasons),
                                          _dp(f64(sigsq)), _dp(f64(final_state)), _dp(out))
        return out

    def ssm_run(self, y, X, observed, prior, opts, spec, rng_setup, init_gamma,
                nsweeps):
        """structural model (f2): spec = structural_spec(...)"""
        T, p = X.shape
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        trend, ns = int(spec["trend"]), int(spec["nseasons"])
        m = self.lib.bo_ssm_create(
            T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), prior["df"], prior["sigma_guess"],
            _dp(f64(prior["pi"])), trend, ns, _dp(f64(spec["var_df"])),
            _dp(f64(spec["var_sigma_guess"])), _dp(f64(spec["var_sigma_upper_limit"])),
            _dp(f64(spec["var_initial_sigma"])), _dp(f64(spec["initial_state_mean"])),
            _dp(f64(spec["initial_state_variance"])))
        ar = spec.get("ar")
        m0 = self.lib.bo_ssm_state_dimension(m)
        if ar:
            rc = self.lib.bo_ssm_add_ar(
                m, int(ar["lags"]), float(ar["df"]), float(ar["sigma_guess"]),
                float(ar["sigma_upper_limit"]), float(ar["initial_sigma"]),
                _dp(f64(ar["initial_phi"])), _dp(f64(spec["initial_state_mean"][m0:])),
                _dp(f64(spec["initial_state_variance"][m0:])))
            assert rc == 0
        dim = self.lib.bo_ssm_state_dimension(m)
        reg = self.lib.bo_ssm_regression(m)
        self.lib.bo_ssvs_set_options(reg, opts["max_model_size"],
                                     opts["sigma_upper_limit"],
                                     opts["swap_threshold"], opts["max_flips"],
                                     opts["draw_beta"], opts["draw_sigma"])
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self.lib.bo_ssvs_set_state(reg, _u8(g0), _dp(np.zeros(p)), 1.0)
        which = [0] + ([1] if trend == 2 else []) + ([2] if ns > 0 else [])
        if rng_setup[0] == "mt":
            glob = self.rng_mt(rng_setup[1])
            # construction order: regression, variance samplers, state
            rngs = ([self.lib.bo_ssvs_rng(reg)]
                    + [self.lib.bo_ssm_variance_rng(m, w) for w in which]
                    + ([self.lib.bo_ssm_ar_rng(m)] if ar else [])
                    + [self.lib.bo_ssm_state_rng(m)])
            for rp in rngs:
                self.lib.bo_rng_seed_mt(rp, self.lib.bo_seed_rng(C.byref(glob)))
            if ar:
                # (ArPosteriorSampler::draw_phi proposes with rmvn_ivar: GlobalRng::rng)
                self.lib.bo_ssm_set_global_rng.argtypes = [C.c_void_p, C.c_void_p]
                self.lib.bo_ssm_set_global_rng(m, C.byref(glob))
        else:
            seed, chain = int(rng_setup[1]), int(rng_setup[2])
            self.lib.bo_rng_seed_philox(self.lib.bo_ssvs_rng(reg), seed, chain, 0, 0)
            for w, sid in ((0, 1), (1, 6), (2, 7)):
                self.lib.bo_rng_seed_philox(self.lib.bo_ssm_variance_rng(m, w), seed,
                                            chain, sid, 0)
            self.lib.bo_rng_seed_philox(self.lib.bo_ssm_state_rng(m), seed, chain, 2, 0)
            self.lib.bo_rng_seed_philox(self.lib.bo_ssm_ar_rng(m), seed, chain, 12, 0)
        nlag = int(ar["lags"]) if ar else 0
        ar_phi = np.zeros((nsweeps, nlag))
        ar_sig = np.zeros(nsweeps)
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        var = np.zeros((nsweeps, 3))
        state = np.zeros((nsweeps, T, dim))
        g = np.zeros(p, dtype=np.uint8)
        b = np.zeros(p)
        s = C.c_double()
        status = 0
        for i in range(nsweeps):
            status = self.lib.bo_ssm_draw(m)
            if status:
                break
            self.lib.bo_ssvs_get_state(reg, _u8(g), _dp(b), C.byref(s))
            gam[i] = g
            beta[i] = b
            sig[i] = s.value
            self.lib.bo_ssm_get_variances(m, _dp(var[i]))
            state[i] = np.ctypeslib.as_array(self.lib.bo_ssm_state(m), (T, dim))
            if ar:
                sa = C.c_double()
                self.lib.bo_ssm_get_ar(m, _dp(ar_phi[i]), C.byref(sa))
                ar_sig[i] = sa.value
        self.lib.bo_ssm_destroy(m)
        return dict(gamma=gam, beta=beta, sigsq=sig, variances=var, state=state,
                    status=status, ar_phi=ar_phi, ar_sigsq=ar_sig)

    def _ssg_build(self, y, X, observed, prior, blocks):
        from cases import general_arrays
        T, p = X.shape
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        L = self.lib
        L.bo_ssm_create_empty.restype = C.c_void_p
        L.bo_ssm_create_empty.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_u8_p,
                                          c_double_p, c_double_p, C.c_double, C.c_double,
                                          c_double_p]
        L.bo_ssm_add_block.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)] + [c_double_p] * 7
        L.bo_ssm_block_rng.restype = C.c_void_p
        L.bo_ssm_block_rng.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.bo_ssm_block_stream_id.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.bo_ssm_block_get.argtypes = [C.c_void_p, C.c_int] + [c_double_p] * 4
        L.bo_ssm_block_set_sigsq.argtypes = [C.c_void_p, C.c_int, c_double_p]
        L.bo_ssm_block_get_ar_suf.argtypes = [C.c_void_p, C.c_int] + [c_double_p] * 4
        m = L.bo_ssm_create_empty(T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs),
                                  _dp(f64(prior["b"])), _dp(fcol(prior["ominv"])),
                                  prior["df"], prior["sigma_guess"], _dp(f64(prior["pi"])))
        kinds, ip, vpar, phi0, a0, P0 = general_arrays(blocks)
        first = 0
        for i, b in enumerate(blocks):
            ipi = np.ascontiguousarray(ip[i], np.int32)
            vp = np.ascontiguousarray(vpar[i].T)   # rows: df, guess, upper limit, initial sigma
            rc = L.bo_ssm_add_block(
                m, int(kinds[i]), ipi.ctypes.data_as(C.POINTER(C.c_int)), _dp(vp[0]), _dp(vp[1]),
                _dp(vp[2]), _dp(vp[3]),
                _dp(f64(b["rotations"] if b["kind"] == 6 else (b["slope_priors"] if b["kind"] == 7 else phi0[i]))),
                _dp(f64(a0[first:first + b["dim"]])),
                _dp(f64(P0[first:first + b["dim"]])))
            assert rc == 0, rc
            first += b["dim"]
        return m

    def ssg_run(self, y, X, observed, prior, opts, blocks, rng_setup, init_gamma, nsweeps,
                state_every=1, set_sigsq=None):
        """general structural model: blocks = cases.general_spec(...).  Returns per sweep
        gamma, beta, sigsq, variances (nblocks x 2), phi (nblocks x 16), the ArModel /
        variance sufficient statistics of the LAST sweep, and the state draws of the
        sweeps i with i % state_every == state_every - 1."""
        T, p = X.shape
        L = self.lib
        m = self._ssg_build(y, X, observed, prior, blocks)
        nb = len(blocks)
        dim = L.bo_ssm_state_dimension(m)
        reg = L.bo_ssm_regression(m)
        L.bo_ssvs_set_options(reg, opts["max_model_size"], opts["sigma_upper_limit"],
                              opts["swap_threshold"], opts["max_flips"], opts["draw_beta"],
                              opts["draw_sigma"])
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        L.bo_ssvs_set_state(reg, _u8(g0), _dp(np.zeros(p)), 1.0)
        nvar = [2 if b["kind"] in (2, 7) else (0 if b["kind"] == 5 else 1) for b in blocks]
        if rng_setup[0] == "mt":
            glob = self.rng_mt(rng_setup[1])
            # construction order: regression, every state model's samplers, state
            rngs = [L.bo_ssvs_rng(reg)]
            for b in range(nb):
                rngs += [L.bo_ssm_block_rng(m, b, v) for v in range(nvar[b])]
            rngs.append(L.bo_ssm_state_rng(m))
            for rp in rngs:
                L.bo_rng_seed_mt(C.c_void_p(rp) if isinstance(rp, int) else rp,
                                 L.bo_seed_rng(C.byref(glob)))
            # (ArPosteriorSampler::draw_phi proposes with rmvn_ivar: GlobalRng::rng)
            L.bo_ssm_set_global_rng.argtypes = [C.c_void_p, C.c_void_p]
            L.bo_ssm_set_global_rng(m, C.byref(glob))
        else:
            seed, chain = int(rng_setup[1]), int(rng_setup[2])
            L.bo_rng_seed_philox(L.bo_ssvs_rng(reg), seed, chain, 0, 0)
            for b in range(nb):
                for v in range(nvar[b]):
                    L.bo_rng_seed_philox(C.c_void_p(L.bo_ssm_block_rng(m, b, v)), seed, chain,
                                         L.bo_ssm_block_stream_id(m, b, v), 0)
            L.bo_rng_seed_philox(L.bo_ssm_state_rng(m), seed, chain, 2, 0)
        if set_sigsq is not None:
            for b in range(nb):
                L.bo_ssm_block_set_sigsq(m, b, _dp(f64(set_sigsq[b])))
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        var = np.zeros((nsweeps, nb, 2))
        phi = np.zeros((nsweeps, nb, 16))
        nkeep = nsweeps // state_every if state_every > 0 else 0
        state = np.zeros((nkeep, T, dim))
        g = np.zeros(p, dtype=np.uint8)
        bb = np.zeros(p)
        s = C.c_double()
        status = 0
        kept = 0
        for i in range(nsweeps):
            status = L.bo_ssm_draw(m)
            if status:
                break
            L.bo_ssvs_get_state(reg, _u8(g), _dp(bb), C.byref(s))
            gam[i] = g
            beta[i] = bb
            sig[i] = s.value
            for b in range(nb):
                L.bo_ssm_block_get(m, b, _dp(var[i, b]), None, None, _dp(phi[i, b]))
            if state_every > 0 and i % state_every == state_every - 1:
                state[kept] = np.ctypeslib.as_array(L.bo_ssm_state(m), (T, dim))
                kept += 1
        suf_n = np.zeros((nb, 2))
        suf_ss = np.zeros((nb, 2))
        ar_suf = {}
        for b in range(nb):
            L.bo_ssm_block_get(m, b, None, _dp(suf_n[b]), _dp(suf_ss[b]), None)
            if blocks[b]["kind"] == 4:
                Lg = blocks[b]["lags"]
                xtx = np.zeros((Lg, Lg)); xty = np.zeros(Lg)
                yty = C.c_double(); n = C.c_double()
                L.bo_ssm_block_get_ar_suf.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p,
                                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]
                L.bo_ssm_block_get_ar_suf(m, b, _dp(xtx), _dp(xty), C.byref(yty), C.byref(n))
                ar_suf[b] = dict(xtx=xtx, xty=xty, yty=yty.value, n=n.value)
        L.bo_ssm_destroy(m)
        return dict(gamma=gam, beta=beta, sigsq=sig, variances=var, phi=phi, state=state,
                    status=status, suf_n=suf_n, suf_ss=suf_ss, ar_suf=ar_suf)

    def ssg_forecast(self, rng, T, newX, beta, sigsq_obs, blocks, sigsq, phi, final_state):
        """simulate_forecast of the general model at fixed parameters: sigsq (nblocks x 2),
        phi (nblocks x 16)"""
        h, p = newX.shape
        prior = dict(b=np.zeros(p), ominv=np.eye(p), df=1.0, sigma_guess=1.0,
                     pi=np.full(p, 0.5))
        blocks = [dict(b) for b in blocks]
        for i, b in enumerate(blocks):
            if b["kind"] == 4:
                b["initial_phi"] = np.asarray(phi[i][:b["lags"]], float)
        m = self._ssg_build(np.zeros(T), np.zeros((T, p)), None, prior, blocks)
        for b in range(len(blocks)):
            self.lib.bo_ssm_block_set_sigsq(m, b, _dp(f64(sigsq[b])))
        out = np.zeros(h)
        self.lib.bo_ssm_forecast_model.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                   c_double_p, c_double_p, C.c_double, c_double_p,
                                                   c_double_p]
        self.lib.bo_ssm_forecast_model(m, C.byref(rng), h, p, _dp(fcol(newX)), _dp(f64(beta)),
                                       float(sigsq_obs), _dp(f64(final_state)), _dp(out))
        self.lib.bo_ssm_destroy(m)
        return out

    def ss_impute_state(self, y, X, observed, beta, gamma, sigsq_obs,
                        sigsq_level, a0, P0, rng):
        T, p = X.shape
        prior = dict(b=np.zeros(p), ominv=np.eye(p), df=1.0, sigma_guess=1.0,
                     pi=np.full(p, 0.5))
        ss = dict(level_df=1.0, level_sigma_guess=1.0,
                  level_sigma_upper_limit=float("inf"), initial_state_mean=a0,
                  initial_state_variance=P0,
                  initial_level_sigma=float(np.sqrt(sigsq_level)))
        m = self.ss_create(y, X, observed, prior, ss)
        self.lib.bo_ss_set_level_sigsq(m, float(sigsq_level))
        reg = self.lib.bo_ss_regression(m)
        g = np.ascontiguousarray(gamma, dtype=np.uint8)
        bb = f64(beta) * g
        self.lib.bo_ssvs_set_state(reg, _u8(g), _dp(bb), float(sigsq_obs))
        st = self.lib.bo_ss_impute_state(m, C.byref(rng))
        assert st == 0
        state = np.ctypeslib.as_array(self.lib.bo_ss_state(m), (T,)).copy()
        n = C.c_double()
        ssq = C.c_double()
        self.lib.bo_ss_level_suf(m, C.byref(n), C.byref(ssq))
        # regression suf lives in the bo_ssvs; read back through a tiny probe:
        out = dict(state=state, level_n=n.value, level_sumsq=ssq.value)
        self.lib.bo_ss_destroy(m)
        return out


    def ss_forecast(self, rng, newX, beta, sigsq_obs, sigsq_level, final_state):
        h, p = newX.shape
        out = np.zeros(h)
        self.lib.bo_ss_simulate_forecast.argtypes = [
            C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p, C.c_double,
            C.c_double, C.c_double, c_double_p]
        self.lib.bo_ss_simulate_forecast(C.byref(rng), h, p, _dp(fcol(newX)), _dp(f64(beta)),
                                         float(sigsq_obs), float(sigsq_level),
                                         float(final_state), _dp(out))
        return out


# ---------------------------------------------------------------------------
def have_ref():
    return os.path.exists(REF_SO)


class Ref:
    """The compiled, unmodified reference (build container only)."""

    def __init__(self):
        self.lib = L = C.CDLL(REF_SO)
        L.ref_last_error.restype = C.c_char_p

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.ref_last_error().decode())

    def uniforms(self, seed, n):
        out = np.zeros(n)
        self._check(self.lib.ref_rng_uniform(C.c_uint64(seed), n, _dp(out)))
        return out

    def seed_rngs(self, seed, n):
        out = np.zeros(n, dtype=np.uint64)
        self._check(self.lib.ref_seed_rng(C.c_uint64(seed), n,
                                          out.ctypes.data_as(C.c_void_p)))
        return out

    def norms(self, seed, n):
        out = np.zeros(n)
        self._check(self.lib.ref_rng_norm(C.c_uint64(seed), n, _dp(out)))
        return out

    def exps(self, seed, n):
        out = np.zeros(n)
        self._check(self.lib.ref_rng_exp(C.c_uint64(seed), n, _dp(out)))
        return out

    def gammas(self, seed, a, b, n):
        out = np.zeros(n)
        self._check(self.lib.ref_rng_gamma(C.c_uint64(seed), C.c_double(a),
                                           C.c_double(b), n, _dp(out)))
        return out

    def trun_gammas(self, seed, a, b, cut, n):
        out = np.zeros(n)
        self._check(self.lib.ref_rng_trun_gamma(C.c_uint64(seed), C.c_double(a),
                                                C.c_double(b), C.c_double(cut),
                                                n, _dp(out)))
        return out

    def random_ints(self, seed, lo, hi, n):
        out = np.zeros(n, dtype=np.int32)
        self._check(self.lib.ref_rng_random_int(C.c_uint64(seed), lo, hi, n,
                                                _ip(out)))
        return out

    def shuffles(self, seed, p, nrep):
        out = np.zeros((nrep, p), dtype=np.int32)
        self._check(self.lib.ref_rng_shuffle(C.c_uint64(seed), p, nrep,
                                             _ip(out)))
        return out

    def rmultis(self, seed, prob, n):
        prob = f64(prob)
        out = np.zeros(n, dtype=np.int32)
        self._check(self.lib.ref_rng_rmulti(C.c_uint64(seed), len(prob),
                                            _dp(prob), n, _ip(out)))
        return out

    def chol(self, A):
        n = A.shape[0]
        L = np.zeros(n * n)
        ok = C.c_int()
        self._check(self.lib.ref_spd_chol(n, _dp(fcol(A)), _dp(L), C.byref(ok)))
        return L.reshape(n, n).T.copy(), bool(ok.value)

    def logdet(self, A):
        out = C.c_double()
        ok = C.c_int()
        self._check(self.lib.ref_spd_logdet(A.shape[0], _dp(fcol(A)),
                                            C.byref(out), C.byref(ok)))
        return out.value, bool(ok.value)

    def solve(self, A, rhs):
        n = A.shape[0]
        x = np.zeros(n)
        ok = C.c_int()
        self._check(self.lib.ref_spd_solve(n, _dp(fcol(A)), _dp(f64(rhs)),
                                           _dp(x), C.byref(ok)))
        return x, bool(ok.value)

    def mdist(self, A, x):
        out = C.c_double()
        self._check(self.lib.ref_spd_mdist(A.shape[0], _dp(fcol(A)),
                                           _dp(f64(x)), C.byref(out)))
        return out.value

    def neregsuf(self, X, y):
        n, p = X.shape
        xtx = np.zeros(p * p)
        xty = np.zeros(p)
        yty = C.c_double()
        ybar = C.c_double()
        xbar = np.zeros(p)
        self._check(self.lib.ref_neregsuf(n, p, _dp(fcol(X)), _dp(f64(y)),
                                          _dp(xtx), _dp(xty), C.byref(yty),
                                          C.byref(ybar), _dp(xbar)))
        return dict(xtx=xtx.reshape(p, p).T.copy(), xty=xty, yty=yty.value,
                    n=float(n), sumy=ybar.value * n, xsum=xbar * n,
                    ybar=ybar.value, xbar=xbar)

    def _opts(self, opts):
        return RefSsvsOptions(**opts)

    def ssvs_run(self, X, y, suf, prior, opts, seed, init_gamma, nsweeps):
        """X,y given -> data path; else sufficient-statistics path."""
        if X is not None:
            n, p = X.shape
        else:
            n, p = int(suf["n"]), len(suf["xty"])
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        o = self._opts(opts)
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        if X is not None:
            args = (_dp(fcol(X)), _dp(f64(y)), None, None, C.c_double(0),
                    C.c_double(0), None)
        else:
            args = (None, None, _dp(fcol(suf["xtx"])), _dp(f64(suf["xty"])),
                    C.c_double(suf["yty"]), C.c_double(suf["sumy"] / suf["n"]),
                    _dp(f64(suf["xsum"] / suf["n"])))
        self._check(self.lib.ref_ssvs_run(
            n, p, *args, _dp(f64(prior["b"])), _dp(fcol(prior["ominv"])),
            C.c_double(prior["df"]), C.c_double(prior["sigma_guess"]),
            _dp(f64(prior["pi"])), C.byref(o), C.c_uint64(seed), _u8(g0),
            nsweeps, _u8(gam), _dp(beta), _dp(sig)))
        return dict(gamma=gam, beta=beta, sigsq=sig)

    def ssvs_run_ctor(self, which, X, y, args5, flag, opts, seed, init_gamma,
                      nsweeps):
        n, p = X.shape
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        o = self._opts(opts)
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        a = [C.c_double(v) for v in args5]
        self._check(self.lib.ref_ssvs_run_ctor(
            which, n, p, _dp(fcol(X)), _dp(f64(y)), *a, int(flag), C.byref(o),
            C.c_uint64(seed), _u8(g0), nsweeps, _u8(gam), _dp(beta), _dp(sig)))
        return dict(gamma=gam, beta=beta, sigsq=sig)

    def adaptive_run(self, suf, prior, opts, seed, init_gamma, nsweeps,
                     max_flips=-1, step_size=-1.0, target=-1.0):
        p = len(suf["xty"])
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self._check(self.lib.ref_adaptive_run(
            int(suf["n"]), p, _dp(fcol(suf["xtx"])), _dp(f64(suf["xty"])),
            C.c_double(suf["yty"]), C.c_double(suf["sumy"] / suf["n"]),
            _dp(f64(suf["xsum"] / suf["n"])), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), C.c_double(prior["df"]),
            C.c_double(prior["sigma_guess"]), _dp(f64(prior["pi"])),
            C.c_int64(opts["max_model_size"]), C.c_double(opts["sigma_upper_limit"]),
            int(max_flips), C.c_double(step_size), C.c_double(target),
            C.c_uint64(seed), _u8(g0), nsweeps, _u8(gam), _dp(beta), _dp(sig)))
        return dict(gamma=gam, beta=beta, sigsq=sig)

    def log_model_prob(self, suf, prior, gammas, max_model_size=-1):
        p = len(suf["xty"])
        G = np.ascontiguousarray(gammas, dtype=np.uint8)
        out = np.zeros(len(G))
        self._check(self.lib.ref_ssvs_log_model_prob(
            int(suf["n"]), p, _dp(fcol(suf["xtx"])), _dp(f64(suf["xty"])),
            C.c_double(suf["yty"]), C.c_double(suf["sumy"] / suf["n"]),
            _dp(f64(suf["xsum"] / suf["n"])), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), C.c_double(prior["df"]),
            C.c_double(prior["sigma_guess"]), _dp(f64(prior["pi"])),
            C.c_int64(max_model_size), len(G), _u8(G), _dp(out)))
        return out

    def logpri(self, suf, prior, gammas, betas, sigsqs, max_model_size=-1):
        p = len(suf["xty"])
        G = np.ascontiguousarray(gammas, dtype=np.uint8)
        B = np.ascontiguousarray(betas, dtype=np.float64)
        S = np.ascontiguousarray(sigsqs, dtype=np.float64)
        out = np.zeros(len(G))
        self._check(self.lib.ref_ssvs_logpri(
            int(suf["n"]), p, _dp(fcol(suf["xtx"])), _dp(f64(suf["xty"])),
            C.c_double(suf["yty"]), C.c_double(suf["sumy"] / suf["n"]),
            _dp(f64(suf["xsum"] / suf["n"])), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), C.c_double(prior["df"]),
            C.c_double(prior["sigma_guess"]), _dp(f64(prior["pi"])),
            C.c_int64(max_model_size), len(G), _u8(G), _dp(B), _dp(S), _dp(out)))
        return out

    def sss_run(self, X, y, w, slab_kind, mu, prec, pi, seed, init_gamma,
                sigsq_seq, max_model_size=-1, max_flips=-1):
        n, p = X.shape
        ns = len(sigsq_seq)
        gam = np.zeros((ns, p), dtype=np.uint8)
        beta = np.zeros((ns, p))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        ww = None if w is None else f64(w)
        self._check(self.lib.ref_sss_run(
            n, p, _dp(fcol(X)), _dp(f64(y)), _dp(ww), int(slab_kind), _dp(f64(mu)),
            _dp(fcol(prec)), _dp(f64(pi)), C.c_int64(max_model_size), int(max_flips),
            C.c_uint64(seed), _u8(g0), ns, _dp(f64(sigsq_seq)), _u8(gam), _dp(beta)))
        return dict(gamma=gam, beta=beta)

    def weighted_suf(self, X, y, w):
        n, p = X.shape
        xtx = np.zeros(p * p)
        xty = np.zeros(p)
        ww = None if w is None else f64(w)
        self._check(self.lib.ref_weighted_suf(n, p, _dp(fcol(X)), _dp(f64(y)), _dp(ww),
                                              _dp(xtx), _dp(xty)))
        return xtx.reshape(p, p).T.copy(), xty

    def ss_run(self, y, X, observed, prior, opts, ss, seed, init_gamma,
               nsweeps):
        T, p = X.shape
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        lev = np.zeros(nsweeps)
        state = np.zeros((nsweeps, T))
        o = self._opts(opts)
        so = RefSsOptions(**ss)
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self._check(self.lib.ref_ss_run(
            T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), C.c_double(prior["df"]),
            C.c_double(prior["sigma_guess"]), _dp(f64(prior["pi"])),
            C.byref(o), C.byref(so), C.c_uint64(seed), _u8(g0), nsweeps,
            _u8(gam), _dp(beta), _dp(sig), _dp(lev), _dp(state)))
        return dict(gamma=gam, beta=beta, sigsq=sig, level_sigsq=lev,
                    state=state)

    def trun_norms(self, seed, mu, sigma, cut, above, n):
        out = np.zeros(n)
        self._check(self.lib.ref_rng_trun_norm(C.c_uint64(seed), C.c_double(mu), C.c_double(sigma),
                                               C.c_double(cut), int(above), n, _dp(out)))
        return out

    def probit_run(self, X, y, ntrials, slab, pi, seed, init_gamma, init_beta, nsweeps,
                   clt_threshold=5, max_model_size=-1, max_flips=-1):
        n, p = X.shape
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self._check(self.lib.ref_probit_run(
            n, p, _dp(fcol(X)), _dp(f64(y)), _dp(f64(ntrials)), _dp(f64(slab["mu"])),
            _dp(fcol(slab["prec"])), _dp(f64(pi)), C.c_int64(max_model_size), int(max_flips),
            int(clt_threshold), C.c_uint64(seed), _u8(g0), _dp(f64(init_beta)), nsweeps,
            _u8(gam), _dp(beta)))
        return dict(gamma=gam, beta=beta)

    def logit_run(self, X, y, ntrials, slab, pi, seed, init_gamma, init_beta, nsweeps,
                  clt_threshold=5, max_model_size=-1, max_flips=-1):
        n, p = X.shape
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self._check(self.lib.ref_logit_run(
            n, p, _dp(fcol(X)), _dp(f64(y)), _dp(f64(ntrials)), _dp(f64(slab["mu"])),
            _dp(fcol(slab["prec"])), _dp(f64(pi)), C.c_int64(max_model_size), int(max_flips),
            int(clt_threshold), C.c_uint64(seed), _u8(g0), _dp(f64(init_beta)), nsweeps,
            _u8(gam), _dp(beta)))
        return dict(gamma=gam, beta=beta)

    def poisson_mixtures(self, y, max_comp=16):
        """the reference table's normal-mixture approximation of NegLogGamma(n) for 1 and
        every positive count in y, asked for in the order a sampler's pass over the data
        asks (the table refits and grows on demand: see ref_poisson_mixtures); packed
        arrays as the oracle / the C-ABI take them"""
        req = []
        for v in np.asarray(y):
            req.append(1)
            if v > 0:
                req.append(int(v))
        req = np.ascontiguousarray(req, dtype=np.int64)
        counts = np.ascontiguousarray(sorted(set(int(c) for c in req)), dtype=np.int64)
        nc = np.zeros(len(counts), np.int32)
        mu = np.zeros((len(counts), max_comp))
        sg = np.zeros((len(counts), max_comp))
        wt = np.zeros((len(counts), max_comp))
        li = C.c_int64()
        self._check(self.lib.ref_poisson_mixtures(
            len(req), req.ctypes.data_as(C.POINTER(C.c_int64)),
            len(counts), counts.ctypes.data_as(C.POINTER(C.c_int64)), max_comp,
            nc.ctypes.data_as(C.POINTER(C.c_int)), _dp(mu), _dp(sg), _dp(wt), C.byref(li)))
        keep = nc > 0
        pack = lambda a: np.concatenate([a[i, :nc[i]] for i in range(len(counts)) if keep[i]])  # noqa: E731
        return dict(counts=counts[keep], ncomp=nc[keep], mu=pack(mu), sigma=pack(sg), weight=pack(wt),
                    largest_index=li.value)

    def poisson_run(self, X, y, exposure, slab, pi, seed, init_gamma, init_beta, nsweeps,
                    max_model_size=-1, max_flips=-1):
        n, p = X.shape
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self._check(self.lib.ref_poisson_run(
            n, p, _dp(fcol(X)), _dp(f64(y)), _dp(f64(exposure)), _dp(f64(slab["mu"])),
            _dp(fcol(slab["prec"])), _dp(f64(pi)), C.c_int64(max_model_size), int(max_flips),
            C.c_uint64(seed), _u8(g0), _dp(f64(init_beta)), nsweeps, _u8(gam), _dp(beta)))
        return dict(gamma=gam, beta=beta)

    def ssm_forecast(self, y, X, beta, gamma, sigsq_obs, trend, nseasons, sigsq, final_state,
                     newX, seed):
        T, p = X.shape
        h = newX.shape[0]
        out = np.zeros(h)
        g = np.ascontiguousarray(gamma, dtype=np.uint8)
        self._check(self.lib.ref_ssm_forecast(
            T, p, _dp(f64(y)), _dp(fcol(X)), _dp(f64(beta)), _u8(g), C.c_double(sigsq_obs),
            int(trend), int(nseasons), _dp(f64(sigsq)), _dp(f64(final_state)), h,
            _dp(fcol(newX)), C.c_uint64(seed), _dp(out)))
        return out

    def ssm_run(self, y, X, observed, prior, opts, spec, seed, init_gamma, nsweeps):
        T, p = X.shape
        trend, ns = int(spec["trend"]), int(spec["nseasons"])
        dim = trend + (ns - 1 if ns > 0 else 0)
        if spec.get("ar"):
            return self._ssm_ar_run(y, X, observed, prior, opts, spec, seed, init_gamma,
                                    nsweeps)
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        var = np.zeros((nsweeps, 3))
        state = np.zeros((nsweeps, T, dim))
        o = self._opts(opts)
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        self._check(self.lib.ref_ssm_run(
            T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), C.c_double(prior["df"]),
            C.c_double(prior["sigma_guess"]), _dp(f64(prior["pi"])), C.byref(o),
            trend, ns, _dp(f64(spec["var_df"])), _dp(f64(spec["var_sigma_guess"])),
            _dp(f64(spec["var_sigma_upper_limit"])), _dp(f64(spec["var_initial_sigma"])),
            _dp(f64(spec["initial_state_mean"])), _dp(f64(spec["initial_state_variance"])),
            C.c_uint64(seed), _u8(g0), nsweeps, _u8(gam), _dp(beta), _dp(sig),
            _dp(var), _dp(state)))
        return dict(gamma=gam, beta=beta, sigsq=sig, variances=var, state=state)

    def _ssm_ar_run(self, y, X, observed, prior, opts, spec, seed, init_gamma, nsweeps):
        T, p = X.shape
        trend, ns, ar = int(spec["trend"]), int(spec["nseasons"]), spec["ar"]
        L = int(ar["lags"])
        dim = trend + (ns - 1 if ns > 0 else 0) + L
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        var = np.zeros((nsweeps, 3))
        state = np.zeros((nsweeps, T, dim))
        out_ar = np.zeros((nsweeps, L + 1))
        o = self._opts(opts)
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        arv = f64([ar["df"], ar["sigma_guess"], ar["sigma_upper_limit"], ar["initial_sigma"]])
        self._check(self.lib.ref_ssm_ar_run(
            T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), C.c_double(prior["df"]),
            C.c_double(prior["sigma_guess"]), _dp(f64(prior["pi"])), C.byref(o),
            trend, ns, _dp(f64(spec["var_df"])), _dp(f64(spec["var_sigma_guess"])),
            _dp(f64(spec["var_sigma_upper_limit"])), _dp(f64(spec["var_initial_sigma"])),
            _dp(f64(spec["initial_state_mean"])), _dp(f64(spec["initial_state_variance"])),
            L, _dp(arv), _dp(f64(ar["initial_phi"])),
            C.c_uint64(seed), _u8(g0), nsweeps, _u8(gam), _dp(beta), _dp(sig),
            _dp(var), _dp(state), _dp(out_ar)))
        return dict(gamma=gam, beta=beta, sigsq=sig, variances=var, state=state,
                    ar_phi=out_ar[:, :L].copy(), ar_sigsq=out_ar[:, L].copy())

    def ssg_run(self, y, X, observed, prior, opts, blocks, seed, init_gamma, nsweeps,
                state_every=1):
        """the compiled reference on a general list of state models (ref_ssg_run)"""
        from cases import general_arrays
        T, p = X.shape
        nb = len(blocks)
        kinds, ip, vpar, phi0, a0, P0 = general_arrays(blocks)
        dim = len(a0)
        gam = np.zeros((nsweeps, p), dtype=np.uint8)
        beta = np.zeros((nsweeps, p))
        sig = np.zeros(nsweeps)
        var = np.zeros((nsweeps, nb, 2))
        phi = np.zeros((nsweeps, nb, 16))
        nkeep = nsweeps // state_every if state_every > 0 else 0
        state = np.zeros((max(nkeep, 1), T, dim))
        o = self._opts(opts)
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        g0 = np.ascontiguousarray(init_gamma, dtype=np.uint8)
        ipc = np.ascontiguousarray(ip, np.int32)
        self._check(self.lib.ref_ssg_run(
            T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs), _dp(f64(prior["b"])),
            _dp(fcol(prior["ominv"])), C.c_double(prior["df"]),
            C.c_double(prior["sigma_guess"]), _dp(f64(prior["pi"])), C.byref(o), nb,
            kinds.ctypes.data_as(C.POINTER(C.c_int)), ipc.ctypes.data_as(C.POINTER(C.c_int)),
            _dp(f64(vpar)), _dp(f64(phi0)), _dp(f64(a0)), _dp(f64(P0)), C.c_uint64(seed),
            _u8(g0), nsweeps, int(state_every), _u8(gam), _dp(beta), _dp(sig), _dp(var),
            _dp(phi), _dp(state)))
        return dict(gamma=gam, beta=beta, sigsq=sig, variances=var, phi=phi,
                    state=state[:nkeep])

    def ssg_forecast(self, T, newX, beta, sigsq_obs, blocks, sigsq, phi, final_state, seed):
        from cases import general_arrays
        h, p = newX.shape
        kinds, ip, _, _, _, _ = general_arrays(blocks)
        out = np.zeros(h)
        g = (np.asarray(beta) != 0).astype(np.uint8)
        ipc = np.ascontiguousarray(ip, np.int32)
        self._check(self.lib.ref_ssg_forecast(
            T, p, _dp(np.zeros(T)), _dp(np.zeros((T, p))), _dp(f64(beta)), _u8(g),
            C.c_double(sigsq_obs), len(blocks), kinds.ctypes.data_as(C.POINTER(C.c_int)),
            ipc.ctypes.data_as(C.POINTER(C.c_int)), _dp(f64(sigsq)), _dp(f64(phi)),
            _dp(f64(final_state)), h, _dp(fcol(newX)), C.c_uint64(seed), _dp(out)))
        return out

    def ss_forecast(self, y, X, beta, gamma, sigsq_obs, sigsq_level, final_state,
                    newX, seed):
        T, p = X.shape
        h = newX.shape[0]
        out = np.zeros(h)
        g = np.ascontiguousarray(gamma, dtype=np.uint8)
        self._check(self.lib.ref_ss_forecast(
            T, p, _dp(f64(y)), _dp(fcol(X)), _dp(f64(beta)), _u8(g),
            C.c_double(sigsq_obs), C.c_double(sigsq_level), C.c_double(final_state),
            h, _dp(fcol(newX)), C.c_uint64(seed), _dp(out)))
        return out

    def ss_impute_state(self, y, X, observed, beta, gamma, sigsq_obs,
                        sigsq_level, a0, P0, seed):
        T, p = X.shape
        state = np.zeros(T)
        xty = np.zeros(p)
        yty = C.c_double()
        n = C.c_double()
        ls = C.c_double()
        ln = C.c_double()
        obs = (None if observed is None
               else np.ascontiguousarray(observed, dtype=np.uint8))
        g = np.ascontiguousarray(gamma, dtype=np.uint8)
        self._check(self.lib.ref_ss_impute_state(
            T, p, _dp(f64(y)), _dp(fcol(X)), _u8(obs), _dp(f64(beta)), _u8(g),
            C.c_double(sigsq_obs), C.c_double(sigsq_level), C.c_double(a0),
            C.c_double(P0), C.c_uint64(seed), _dp(state), _dp(xty),
            C.byref(yty), C.byref(n), C.byref(ls), C.byref(ln)))
        return dict(state=state, xty=xty, yty=yty.value, n=n.value,
                    level_sumsq=ls.value, level_n=ln.value)
