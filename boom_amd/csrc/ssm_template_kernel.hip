// bsts structural time series for many chains: the state half of
// StateSpacePosteriorSampler::draw() when the state is a trend block
// (LocalLevelStateModel, or LocalLinearTrendStateModel with one
// ZeroMeanMvnIndependenceSampler per variance) plus an optional
// SeasonalStateModel(nseasons, season_duration = 1) and an optional
// ArStateModel(lags) with its ArPosteriorSampler.  SURVEY 8f row f2.
//
//   state model samplers                 (ZeroMeanGaussianConjSampler.cpp:57-60,
//                                         ZeroMeanMvnIndependenceSampler.cpp:63-70)
//   Base::impute_state                   (StateSpaceModelBase.cpp:278-291)
//     ScalarBase::simulate_forward       (:771-790) with
//       ScalarMarginalDistribution::update (ScalarKalmanFilter.cpp:41-83), vector state
//       StateModelBase::simulate_initial_state (StateModel.cpp:47-56)
//       simulate_state_error (LocalLevelStateModel.cpp:62-64, MvnBase.cpp:257,
//                             SeasonalStateModel.cpp:124-146)
//     Base::propagate_disturbances       (:858-891), fast_disturbance_smooth
//                                          (ScalarKalmanFilter.cpp:168-196)
//     observe_state (LocalLevelStateModel.cpp:52-58, LocalLinearTrend.cpp:53-63,
//                    SeasonalStateModel.cpp:74-86, ArStateModel.cpp:64-69),
//     observe_data_given_state
//   ArPosteriorSampler::draw             (ArPosteriorSampler.cpp:52-143)
//
// State vector [trend (1 or 2) | seasonal (nseasons - 1) | autoregression (lags)],
// dimension m <= 16.
//   Z    ones at the first element of each block
//   T    trend [1] or [[1, 1], [0, 1]]; seasonal: first row -1, ones below the diagonal;
//        autoregression: first row phi, ones below the diagonal
//   RQR  diagonal: level, slope and the first element of the seasonal / autoregression block
// One chain per workgroup of two wavefronts: both share the adjusted
// observations and the sweep's normals (stream_normals.h), then wave 0 runs the
// three passes over time.  Lane j < m holds component j of every state-sized
// vector and column j of the state variance P (16 registers).  Unlike the
// local-level kernel (kalman_kernel.hip) the passes are SERIAL in time: the
// per-step maps are m x m here and their compositions no longer fit a wave
// scan.  As there, the data filter and the simulation filter share the gains, so
// ONE filter runs on w = y* - y+, and one smoother on the difference.
#include <hip/hip_runtime.h>

#include "diag.h"
#include "ktimer.h"

#include "device_rng.h"
#include "kalman_params.h"
#include "stream_normals.h"

// (round 4) This is the round-3 kernel, kept as the FAST PATH of the block lists that
// are its template -- [local level | local linear trend] [+ seasonal of duration 1]
// [+ autoregression], state dimension <= 16: its passes are compiled for the shape
// (template flags), where the general kernel (ssm_kernel.hip) walks a block list.  Same
// draws, same storage: Tpl below maps the template's names onto the general layout.

namespace boom_amd {

namespace {

constexpr int WAVE = 64;
constexpr int SSM_MAX = 16;        // the template's state dimension limit (= AR_MAX)
constexpr int PLD = SSM_MAX + 1;   // leading dimension of the state variance in LDS
constexpr int V_FAILED = 1 << 30;  // s_vprog: the variance pass stopped (F <= 0)

// the template's view of the general specification / storage (no arrays: an index the
// compiler does not see as a constant would put them in scratch memory)
struct Tpl {
  const SsgSpec *S;
  const SsmParams &M;
  int m, trend, nseasons, s0, ar_lags, ar0;
  int av;         // variance index of the autoregression's error variance
  // variance index of level (0), slope (1), seasonal (2): 0, 1, trend
  __device__ __forceinline__ int vi(int i) const { return i == 2 ? trend : i; }
  __device__ __forceinline__ size_t at(int chain, int i) const { return (size_t)chain * SSG_MAX_VAR + vi(i); }
  __device__ __forceinline__ size_t ar_at(int chain) const { return (size_t)chain * SSG_MAX_VAR + av; }
  __device__ __forceinline__ double *ar_phi(int chain) const { return M.ar_phi + (size_t)chain * SSG_MAX_AR * AR_MAX; }
  __device__ __forceinline__ double *ar_suf(int chain) const { return M.ar_suf + (size_t)chain * SSG_MAX_AR * AR_SUF_STRIDE; }
};
__device__ __forceinline__ Tpl make_tpl(const SsmParams &M) {
  const int ar0 = M.tpl_trend + (M.tpl_nseasons > 0 ? M.tpl_nseasons - 1 : 0);
  return Tpl{M.spec, M, M.m, M.tpl_trend, M.tpl_nseasons, M.tpl_nseasons > 0 ? M.tpl_trend : -1,
             M.tpl_ar_lags, ar0, M.tpl_trend + (M.tpl_nseasons > 0 ? 1 : 0)};
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double sdpp(double x, double fill) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned long long f = __builtin_bit_cast(unsigned long long, fill);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// value of lane `src` (wave-uniform src)
__device__ __forceinline__ double rl(double x, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum over the first 16 lanes (the vector's lanes; the others hold 0), everywhere
__device__ __forceinline__ double row_total(double x) {
  x += sdpp<0x111, 0xf>(x, 0.0);  // row_shr:1
  x += sdpp<0x112, 0xf>(x, 0.0);
  x += sdpp<0x114, 0xf>(x, 0.0);
  x += sdpp<0x118, 0xf>(x, 0.0);
  return rl(x, 15);
}

// the structure of the transition matrix
struct Shape {
  int m, trend, s0, ns;   // ns: size of the seasonal block (0: none)
  int a0, na;             // the autoregression block: first index, size (0: none)
  __device__ __forceinline__ bool seasonal(int i) const { return ns > 0 && i >= s0 && i < s0 + ns; }
  __device__ __forceinline__ bool ar(int i) const { return na > 0 && i >= a0 && i < a0 + na; }
};

// The seasonal block is kept in a ROTATING layout: logical component i at time
// t (0 = the current season's effect, i = the effect i seasons back) lives in
// physical slot (c_t + i) mod ns, c_t = (-t) mod ns.  The transition
// (new[0] = -sum(old), new[i] = old[i - 1]) then moves nothing: the slot of the
// component that drops out, c_t - 1, receives the new first component and
// becomes c_{t+1}.  For the variance that turns T P T' from O(m^2) data
// movement into one new row / column per step.
__device__ __forceinline__ int cursor_at(int t, int ns) {
  const int r = t % ns;
  return r == 0 ? 0 : ns - r;
}
__device__ __forceinline__ int cursor_prev(int c, int ns) { return c == 0 ? ns - 1 : c - 1; }   // c_{t+1} from c_t

// y = T x for a vector held one component per lane; c: cursor of x's layout (the
// result is in the next step's layout)
// The autoregression block keeps its logical order (lane a0 + i = lag i); phl: this
// lane's coefficient (0 outside the block).
template <int TREND, bool SEAS, bool AR>
__device__ __forceinline__ double vecT(const Shape &S, double x, int lane, int c, double phl) {
  double y = x;
  if (TREND == 2) {
    const double x1 = rl(x, 1);
    if (lane == 0) y = x + x1;
  }
  if (SEAS) {
    const double tot = row_total(S.seasonal(lane) ? x : 0.0);
    if (lane == S.s0 + cursor_prev(c, S.ns)) y = -tot;
  }
  if (AR) {
    // new[0] = phi'old, new[i] = old[i - 1]  (AutoRegressionTransitionMatrix, SparseMatrix.cpp:1261-1310)
    const double tot = row_total(phl * x);
    const double below = sdpp<0x111, 0xf>(x, 0.0);   // row_shr:1
    if (S.ar(lane)) y = (lane == S.a0) ? tot : below;
  }
  return y;
}
// y = T' x; c1: cursor of x's layout (the result is in the previous step's)
template <int TREND, bool SEAS, bool AR>
__device__ __forceinline__ double vecTt(const Shape &S, double x, int lane, int c1, double phl) {
  double y = x;
  if (TREND == 2) {
    const double x0 = rl(x, 0);
    if (lane == 1) y = x0 + x;
  }
  if (SEAS) {
    const double first = rl(x, S.s0 + c1);
    if (S.seasonal(lane)) y = (lane == S.s0 + c1) ? -first : x - first;
  }
  if (AR) {
    // out[i] = phi_i x[0] + x[i + 1]  (Tmult, SparseMatrix.cpp:1286-1295)
    const double first = rl(x, S.a0);
    const double above = sdpp<0x101, 0xf>(x, 0.0);   // row_shl:1
    if (S.ar(lane)) y = phl * first + ((lane + 1 < S.a0 + S.na) ? above : 0.0);
  }
  return y;
}
// Z'x, c: cursor of x's layout
template <bool SEAS, bool AR>
__device__ __forceinline__ double zdot(const Shape &S, double x, int c) {
  double a = rl(x, 0);
  if (SEAS) a += rl(x, S.s0 + c);
  if (AR) a += rl(x, S.a0);
  return a;
}
// a block of `n` doubles between HBM and LDS, by one wave
__device__ __forceinline__ void blk_load(double *lds, const double *g, int n, int lane) {
  for (int i = lane; i < n; i += WAVE) lds[i] = g[i];
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void blk_store(double *g, const double *lds, int n, int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < n; i += WAVE) g[i] = lds[i];
  __builtin_amdgcn_wave_barrier();
}

// ---- ArPosteriorSampler::draw for one chain, by one (whole) wave.  Vectors sit one
// component per lane (lane i < L), the L x L matrices in LDS at leading dimension
// SSM_MAX; every lane reads the same random numbers.
struct ArLds {
  double X[SSM_MAX * SSM_MAX];    // xtx
  double Lc[SSM_MAX * SSM_MAX];   // chol(xtx)
  double Lp[SSM_MAX * SSM_MAX];   // chol(xtx / sigsq)
};
// lower Cholesky factor of `scale` * A (A symmetric, full storage); false: not positive definite
__device__ __forceinline__ bool ar_chol(const double *A, double scale, double *Lc, int L, int lane) {
  for (int j = 0; j < L; ++j) {
    double sacc = 0.0;
    if (lane >= j && lane < L) {
      sacc = A[lane * SSM_MAX + j] * scale;
      for (int k = 0; k < j; ++k) sacc -= Lc[lane * SSM_MAX + k] * Lc[j * SSM_MAX + k];
    }
    const double djj = rl(sacc, j);
    if (!(djj > 0.0)) return false;
    const double d = sqrt(djj);
    if (lane == j) Lc[j * SSM_MAX + j] = d;
    else if (lane > j && lane < L) Lc[lane * SSM_MAX + j] = sacc / d;
    __builtin_amdgcn_wave_barrier();
  }
  return true;
}
// x: Lc x = b
__device__ __forceinline__ double ar_lsolve(const double *Lc, double b, int L, int lane) {
  double x = 0.0;
  for (int i = 0; i < L; ++i) {
    const double tot = row_total((lane < i) ? Lc[i * SSM_MAX + lane] * x : 0.0);
    const double xi = (rl(b, i) - tot) / Lc[i * SSM_MAX + i];
    if (lane == i) x = xi;
  }
  return x;
}
// x: Lc' x = b
__device__ __forceinline__ double ar_ltsolve(const double *Lc, double b, int L, int lane) {
  double x = 0.0;
  for (int i = L - 1; i >= 0; --i) {
    const double tot = row_total((lane > i && lane < L) ? Lc[lane * SSM_MAX + i] * x : 0.0);
    const double xi = (rl(b, i) - tot) / Lc[i * SSM_MAX + i];
    if (lane == i) x = xi;
  }
  return x;
}
// ArModel::check_stationary (ArModel.cpp:142-170).  The quick bound sum |phi| < 1,
// then -- where the reference finds the polynomial's roots (Jenkins-Traub) -- the
// equivalent step-down recursion: every partial autocorrelation inside (-1, 1).
__device__ __forceinline__ bool ar_stationary(double a, int L, int lane) {
  if (row_total((lane < L) ? fabs(a) : 0.0) < 1.0) return true;
  for (int k = L; k >= 1; --k) {
    const double r = rl(a, k - 1);
    if (!(fabs(r) < 1.0)) return false;
    const double den = 1.0 - r * r;
    const int src = k - 2 - lane;
    const double rev = __shfl(a, src < 0 ? 0 : src);
    if (lane < k - 1) a = (a + r * rev) / den;
  }
  return true;
}
// Tn2Sampler (distributions/Tn2Sampler.cpp:25-131): adaptive rejection sampling of a
// standard normal on [lo, hi] under the hull of tangents at the points x (logf =
// -x^2/2).  As in d_ars_gamma_tail the hull lives across the wave: lane i holds
// point i (abscissa, log density, slope, cdf) and knot i; lane n holds the last knot,
// so up to 63 points.  Every lane of the wave must be active.  Restated as written,
// update_cdf's increment (exp(y - y0) / d) * expm1(d * knots[k + 1] - knots[k]) included.
__device__ __forceinline__ double ar_tn2_draw(SeqRng &rng, double lo, double hi, int *bad) {
  const int lane = (int)(threadIdx.x & 63);
  int n = 2;
  double xs = (lane == 0) ? lo : hi;   // (lanes past n - 1 hold copies of the last point)
  double ys = -.5 * xs * xs, ds = -xs, kn = 0.0, cdf = 0.0;
  for (int level = 0; level <= 1001; ++level) {
    // refresh_knots: knots[0] = x[0], knots[n] = x[n - 1], compute_knot in between
    {
      const double x1 = __shfl_up(xs, 1), y1 = __shfl_up(ys, 1), d1 = __shfl_up(ds, 1);
      double ans = (y1 - d1 * x1) - (ys - ds * xs);
      ans /= (ds - d1);
      kn = (lane == 0) ? xs : ((lane >= n) ? x1 : ans);
    }
    // update_cdf
    {
      const double y0 = ars_lane(ys, 0);
      const double knext = __shfl_down(kn, 1);
      const double y = ys + ds * (kn - xs);
      const double inc = (fabs(ds) < .00000000001) ? exp(y - y0) * (knext - kn)
                                                  : (exp(y - y0) / ds) * expm1(ds * knext - kn);
      double last = 0.0;
      for (int k = 0; k < n; ++k) {
        const double ik = ars_lane(inc, k);
        last = (k == 0) ? ik : last + ik;
        if (lane == k) cdf = last;
      }
    }
    const double u = d_runif(rng, 0.0, ars_lane(cdf, n - 1));
    const int k = ars_lower_bound(cdf, n, u);
    if (k >= n) break;   // (past the end of cdf in the reference)
    const double klo = ars_lane(kn, k), khi = ars_lane(kn, k + 1);
    const double dk = ars_lane(ds, k);
    const double lam = -1 * dk;
    double cand;
    if (lam == 0 || fabs(khi - klo) < 1.4901161193847656e-08) cand = d_runif(rng, klo, khi);   // sqrt(epsilon)
    else cand = d_rtrun_exp(rng, lam, klo, khi);
    const double target = -.5 * cand * cand;
    const double logu = (ars_lane(ys, k) + dk * (cand - ars_lane(xs, k))) - d_rexp(rng, 1.0);
    if (logu < target) return cand;
    // add_point (an error in the reference when the candidate left [x[0], x.back()])
    if (cand > ars_lane(xs, n - 1) || cand < ars_lane(xs, 0) || n >= 63) break;
    const int pos = ars_lower_bound(xs, n, cand);
    {
      const double xu = __shfl_up(xs, 1), yu = __shfl_up(ys, 1), du = __shfl_up(ds, 1);
      if (lane > pos) { xs = xu; ys = yu; ds = du; }
      if (lane == pos) { xs = cand; ys = target; ds = -cand; }
    }
    ++n;
  }
  *bad = 1;
  return 0.0;
}
// rtrun_norm_2_mt (trun_norm.cpp:273-325), lo and hi finite: the two rejection samplers
// of lo < mu < hi, the Tn2Sampler in the tails
__device__ __forceinline__ double ar_rtrun_norm_2(SeqRng &rng, double mu, double sigma, double lo, double hi,
                                                  int *bad) {
  if (lo < mu && hi > mu) {
    if ((hi - lo) / sigma > .5) {
      double y = lo - 1;
      while (y < lo || y > hi) y = d_rnorm(rng, mu, sigma);
      return y;
    }
    const double ln_sqrt_2pi = 0.918938533204672741780329736406;
    const double phi_mu = -(ln_sqrt_2pi + 0.5 * 0.0 * 0.0 + log(sigma));
    double phi = phi_mu, u = phi + 1, y = 0;
    while (u > phi) {
      y = d_runif(rng, lo, hi);
      const double x = (y - mu) / sigma;
      phi = -(ln_sqrt_2pi + 0.5 * x * x + log(sigma));
      u = phi_mu - d_rexp(rng, 1.0);
    }
    return y;
  }
  hi = (hi - mu) / sigma;
  lo = (lo - mu) / sigma;
  if (hi < 0) {
    // (the reference recurses with (0, 1, -hi, -lo), which lands in its Tn2Sampler)
    const double y = ar_tn2_draw(rng, -hi, -lo, bad);
    return mu - sigma * y;
  }
  const double y = ar_tn2_draw(rng, lo, hi, bad);
  return y * sigma + mu;
}
// draw_phi (up to three multivariate proposals, else one coefficient at a time) and
// draw_sigma.  phi_l: the lane's coefficient (in: current, out: drawn); *sigsq likewise.
__device__ __forceinline__ int ar_draw(ArLds &W, const Tpl &Q, int chain, SeqRng &rng, double &phi_l,
                                       double &sigsq, int lane) {
  const int L = Q.ar_lags;
  const double *suf = Q.ar_suf(chain);
  for (int e = lane; e < SSM_MAX * SSM_MAX; e += WAVE) W.X[e] = suf[e];
  const double xty = (lane < L) ? suf[AR_SUF_XTY + lane] : 0.0;
  const double yty = suf[AR_SUF_YTY], n = suf[AR_SUF_N];
  __builtin_amdgcn_wave_barrier();
  if (!ar_chol(W.X, 1.0, W.Lc, L, lane)) return CHAIN_NOT_PD;
  const double phi_hat = ar_ltsolve(W.Lc, ar_lsolve(W.Lc, xty, L, lane), L, lane);
  // rmvn_ivar(phi_hat, xtx / sigsq)
  if (!ar_chol(W.X, 1.0 / sigsq, W.Lp, L, lane)) return CHAIN_NOT_PD;
  bool ok = false;
  for (int attempt = 0; attempt < 3 && !ok; ++attempt) {
    double z = 0.0;
    for (int i = 0; i < L; ++i) {
      const double zi = d_rnorm(rng, 0.0, 1.0);
      if (lane == i) z = zi;
    }
    const double zs = ar_ltsolve(W.Lp, z, L, lane);   // (whole wave: the lanes talk to each other)
    const double cand = (lane < L) ? zs + phi_hat : 0.0;
    ok = ar_stationary(cand, L, lane);
    if (ok) phi_l = cand;
  }
  if (!ok) {
    double ph = phi_l;
    if (!ar_stationary(ph, L, lane)) return CHAIN_RNG_BRANCH;
    for (int i = 0; i < L; ++i) {
      const double initial_phi = rl(ph, i);
      double lo = -1, hi = 1;
      const double ivar = W.X[i * SSM_MAX + i];
      const double dot = row_total((lane < L) ? ph * W.X[lane * SSM_MAX + i] : 0.0);
      const double mu = (rl(xty, i) - (dot - initial_phi * ivar)) / ivar;
      for (;;) {
        int bad = 0;
        const double candidate = ar_rtrun_norm_2(rng, mu, sqrt(1.0 / ivar), lo, hi, &bad);
        if (bad) return CHAIN_RNG_BRANCH;
        if (lane == i) ph = candidate;
        if (ar_stationary(ph, L, lane)) break;
        if (candidate > initial_phi) hi = candidate; else lo = candidate;
      }
    }
    phi_l = ph;
  }
  // draw_sigma: ss = phi' xtx phi - 2 phi' xty + yty, df = n
  double row = 0.0;
  for (int j = 0; j < L; ++j) {
    const double pj = rl(phi_l, j);
    if (lane < L) row += W.X[lane * SSM_MAX + j] * pj;
  }
  const double quad = row_total((lane < L) ? phi_l * row : 0.0);
  const double lin = row_total((lane < L) ? phi_l * xty : 0.0);
  const double ss = quad - 2 * lin + yty;
  int bad = 0;
  sigsq = d_draw_variance(rng, n + Q.S->prior_df[Q.av], ss + Q.S->prior_ss[Q.av], Q.S->sigma_max[Q.av], &bad);
  return bad ? CHAIN_RNG_BRANCH : CHAIN_OK;
}

// ---- time across the lanes (round 4).  Two of the five passes have no gain in them: the
// simulation alpha+_{t+1} = T alpha+_t + eta_t and the mean correction m_{t+1} = T m_t +
// RQR r_t are sums of their inputs --
//   slope_t = slope_0 + sum_{s<=t} n1_s,   level_t = level_0 + sum_{s<=t} (slope_{s-1} + n0_s),
//   seasonal, x_t = the block's first component at time t (component i is x_{t-i}):
//     x_t = -(x_{t-1} + ... + x_{t-ns}) + N_t; subtracting the same line for t - 1:
//     x_t = x_{t-ns-1} + (N_t - N_{t-1}) for t >= 2: ns + 1 interleaved running sums --
// so lane l of a block of 64 steps takes time tb + l and the sums are wave scans (six
// ds_bpermute steps; the seasonal one in strides of ns + 1), with the carries of the
// previous block.  250 instructions per 64 steps where the serial passes spent 64 x 90
// (simulation) and 64 x 180 (correction + statistics): the launch's critical path was
// V + F + B + C and is V + B now.  (The autoregression block's recursion has its
// coefficients in it: those shapes keep the serial passes.)  The sums are the
// reference's sums in another order: results agree to rounding (1e-13 observed), inside
// the stated 1e-8.
__device__ __forceinline__ double scan_incl(double x, int lane, int first_shift) {
  for (int d = first_shift; d < WAVE; d <<= 1) {
    const double y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  return x;
}
__device__ __forceinline__ double wave_total(double x) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) x += __shfl_xor(x, d);
  return x;
}
struct PathCarry {
  double slope, level;   // the values at the previous block's last step
  double nprev;          // the seasonal input of the previous block's last step
  double prevx;          // lane l: x at the previous block's step l
};
struct PathStep {
  double lev, slo;
  double seas[SSM_MAX - 1];   // component i of the seasonal block (logical order)
};
// one block: lane = time t (in: t < T); n0, n1, N2: the level's, slope's and seasonal inputs
// of time t (0 for t = 0 and past T); cidx: where lane l finds its predecessor of the
// previous block in the strided sum, 64 - (ns + 1) + l mod (ns + 1)
template <int TREND, bool SEAS>
__device__ __forceinline__ void path_block(const Shape &S, int lane, int cidx, int t, bool in, double n0, double n1,
                                           double N2, PathCarry &C, double *s_xw, PathStep &out) {
  double sl = 0.0;
  if (TREND == 2) sl = C.slope + scan_incl(n1, lane, 1);
  double linc = n0;
  if (TREND == 2) {
    double slm1 = __shfl_up(sl, 1);
    if (lane == 0) slm1 = C.slope;
    linc = (in && t >= 1) ? slm1 + n0 : 0.0;
  }
  const double lv = C.level + scan_incl(linc, lane, 1);
  out.lev = lv;
  out.slo = sl;
  if (SEAS) {
    double np = __shfl_up(N2, 1);
    if (lane == 0) np = C.nprev;
    const double g = (in && t >= 2) ? N2 - np : 0.0;
    const double x = scan_incl(g, lane, S.ns + 1) + __shfl(C.prevx, cidx);
    // the window of the last sixteen values of x in front of the block's own
    if (lane >= WAVE - 16) s_xw[lane - (WAVE - 16)] = C.prevx;
    s_xw[16 + lane] = x;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < SSM_MAX - 1; ++i) out.seas[i] = (i < S.ns) ? s_xw[16 + lane - i] : 0.0;
    wave_lds_sync();
    C.prevx = x;
    C.nprev = rl(N2, WAVE - 1);
  }
  C.slope = rl(sl, WAVE - 1);
  C.level = rl(lv, WAVE - 1);
}
// the carries in front of time 0: init = the vector at time 0, one component per lane
// (cursor 0: physical = logical order); N2_1 = the seasonal input of time 1
template <int TREND, bool SEAS>
__device__ __forceinline__ void path_start(const Shape &S, int lane, double init, double N2_1, PathCarry &C) {
  C.level = rl(init, 0);
  C.slope = (TREND == 2) ? rl(init, 1) : 0.0;
  C.nprev = 0.0;
  C.prevx = 0.0;
  if (SEAS) {
    const int P = S.ns + 1;
    const double x0 = rl(init, S.s0);
    const double x1 = -row_total(S.seasonal(lane) ? init : 0.0) + N2_1;
    // lane 64 - P + rho stands for time rho - P: x_{rho - P} = init[s0 + P - rho] for rho >= 2; the lanes of
    // times -P and -P + 1 carry x_0 and x_1 (which have no predecessor of their own)
    const int rho = lane - (WAVE - P);
    const double v = __shfl(init, (rho >= 2) ? S.s0 + P - rho : 0);
    C.prevx = (rho < 0) ? 0.0 : (rho == 0 ? x0 : (rho == 1 ? x1 : v));
  }
}

// the AR block's first component (one or two lags) over the block's steps: every lane runs the
// same short recursion, lane 0 leaves the values in LDS behind the two before the block
// (s_aw[14], s_aw[15]); n_l: the step's input by lane = time; a1, a2: the values at t - 1, t - 2
__device__ __forceinline__ void ar_block(double *s_aw, int lane, int tb, int nstep, double n_l, double init,
                                         double ph0, double ph1, double &a1, double &a2) {
  if (lane == 0) { s_aw[14] = a2; s_aw[15] = a1; }
#pragma nounroll
  for (int s = 0; s < nstep; ++s) {
    // (the reference: phi . lags, then + the error)
    const double an = (tb + s == 0) ? init : (ph0 * a1 + ph1 * a2) + rl(n_l, s);
    if (lane == 0) s_aw[16 + s] = an;
    a2 = a1;
    a1 = an;
  }
  wave_lds_sync();
}

}  // namespace

// grid = chains, block = 128.  TREND: 1 local level, 2 local linear trend; SEAS: a
// seasonal block follows; AR: an autoregression block follows
template <int TREND, bool SEAS, bool AR>
__global__ __launch_bounds__(128) void ssm_simsmooth_kernel(SsParams P, int draw_variances) {
  // (the passes' buffers take the place of the normals generator's lists, which
  // are done by then: 37 KB per workgroup, four workgroups per CU)
  struct PassLds {
    double blk[2][WAVE * SSM_MAX];   // a block of 64 steps of a state-sized series, per wave
    double P[SSM_MAX * PLD];         // the state variance (wave 1), rows PLD apart
    double tv[SSM_MAX];
  };
  union SharedLds {
    NormalsLds norm;
    PassLds pass;
    ArLds ar;
  };
  __shared__ SharedLds s_lds;
  __shared__ int s_flag;
  __shared__ int s_vprog;                 // blocks of 64 steps the variance pass has put out (wave 1 -> wave 0)
  __shared__ double s_xw[16 + WAVE];        // the seasonal scans' window (wave 0)
  __shared__ double s_aw[16 + WAVE];        // the AR block's first component over a block, behind its predecessors
  __shared__ int s_cprog, s_cdone;        // the last pass: blocks of state draws wave 0 has made / wave 1 has taken
  __shared__ double s_phi[SSM_MAX + 1];   // the autoregression coefficients, then the block's error variance
  double (&s_blk)[2][WAVE * SSM_MAX] = s_lds.pass.blk;
  double (&s_P)[SSM_MAX * PLD] = s_lds.pass.P;
  double (&s_pz)[SSM_MAX] = s_lds.pass.tv;   // PZ of the step, for every lane to read (broadcast)
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) return;
  if (P.only_ran && P.only_ran[chain] == 0) return;
  const Tpl Q = make_tpl(P.ssm);
  const int T = P.T, p = P.p, m = Q.m;
  Shape S;
  S.m = m; S.trend = TREND; S.s0 = TREND; S.ns = SEAS ? Q.nseasons - 1 : 0;
  S.a0 = AR ? Q.ar0 : 0; S.na = AR ? Q.ar_lags : 0;
  // the simulation and the last pass with time across the lanes: every shape but an AR block of
  // more than two lags (its first component is then a short serial recursion per block)
  const bool scans = !AR || S.na <= 2;
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  int status = CHAIN_OK;
  if (threadIdx.x == 0) { s_flag = CHAIN_OK; s_vprog = 0; s_cprog = 0; s_cdone = 0; }
#ifdef BA_KSTAMPS
  long long kph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, klast = (long long)__builtin_readcyclecounter();
#endif

  // ---- the state models' variance draws, in model order: level [, slope], seasonal
  double sig2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) sig2[i] = Q.M.var_sigsq[Q.at(chain, i)];
  if (draw_variances) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool active = (i == 0) || (i == 1 && TREND == 2) || (i == 2 && SEAS);
      if (active) {
        const uint32_t sid = (i == 0) ? 1u : (i == 1 ? 6u : 7u);
        SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, sid}, Q.M.pos_var[Q.at(chain, i)]};
        int bad = 0;
        const double DF = Q.M.var_n[Q.at(chain, i)] + Q.S->prior_df[Q.vi(i)];
        const double SSQ = Q.M.var_ss[Q.at(chain, i)] + Q.S->prior_ss[Q.vi(i)];
        double draw = d_draw_variance(rng, DF, SSQ, Q.S->sigma_max[Q.vi(i)], &bad);
        if (bad) status = CHAIN_RNG_BRANCH;
        // ZeroMeanMvnIndependenceSampler sets siginv(i, i) = 1 / draw; the model's
        // Sigma is the inverse of that again
        if (TREND == 2 && i < 2) draw = 1.0 / (1.0 / draw);
        sig2[i] = draw;
        // (every thread makes the draw from the position it READ: none may find the new one --
        // a wave that fell one draw behind did, under load, until round 4's stress runs)
        __syncthreads();
        if (lane == 0 && wave == 0) {
          Q.M.pos_var[Q.at(chain, i)] = rng.pos;
          Q.M.var_sigsq[Q.at(chain, i)] = draw;
        }
      }
    }
  }
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }
  // ---- the autoregression block's sampler (after the seasonal model's), by wave 0
  double phl = 0.0, sig2a = 0.0;
  if (AR) {
    if (wave == 0) {
      phl = S.ar(lane) ? Q.ar_phi(chain)[lane - S.a0] : 0.0;
      sig2a = Q.M.var_sigsq[Q.ar_at(chain)];
      if (draw_variances) {
        SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 12u}, Q.M.pos_var[Q.ar_at(chain)]};
        // (the sampler's vectors sit at lanes 0 .. L - 1)
        double ph = __shfl(phl, lane + S.a0);
        if (lane >= S.na) ph = 0.0;
        const int st = ar_draw(s_lds.ar, Q, chain, rng, ph, sig2a, lane);
        if (st != CHAIN_OK) {
          if (lane == 0) s_flag = st;
        } else {
          if (lane < S.na) Q.ar_phi(chain)[lane] = ph;
          if (lane == 0) {
            Q.M.var_sigsq[Q.ar_at(chain)] = sig2a;
            Q.M.pos_var[Q.ar_at(chain)] = rng.pos;
          }
        }
        phl = __shfl(ph, lane >= S.a0 ? lane - S.a0 : 0);
        if (!S.ar(lane)) phl = 0.0;
      }
      const double pv = __shfl(phl, (lane + S.a0) & 63);
      if (lane < SSM_MAX) s_phi[lane] = (lane < S.na) ? pv : 0.0;
      if (lane == 0) s_phi[SSM_MAX] = sig2a;
    }
    __syncthreads();
    status = s_flag;
    if (status != CHAIN_OK) {
      if (threadIdx.x == 0) P.status[chain] = status;
      return;
    }
    if (wave == 1) {
      phl = S.ar(lane) ? s_phi[lane - S.a0] : 0.0;
      sig2a = s_phi[SSM_MAX];
    }
  }
  const double sda = sqrt(sig2a);

  const double H = P.sigsq[chain], sqrtH = sqrt(H);
  const double sdv[3] = {sqrt(sig2[0]), sqrt(sig2[1]), sqrt(sig2[2])};
  const double *beta = P.beta + (size_t)chain * p;
  double *w0 = P.scratch + (size_t)chain * P.scratch_stride;   // y* -> w = y* - y+ -> (v - v+) / F
  double *sres = w0 + T;                                       // F_t, then residuals (input of the X'e GEMM)
  double *wk = Q.M.work + (size_t)chain * Q.M.work_stride;
  double *gK = wk;                                 // K_t, m per step (layout of step t + 1)
  double *gst = gK + (size_t)m * T;                // alpha+_t (layout of step t), then the state draw
  double *gd = gst + (size_t)m * T;                // r_t (difference) at the four rows with state error: 4 series of T
  double *szz = gd + (size_t)4 * T;                // the sweep's normals

  SSTAMP(0);
  // ---- 1. adjusted observations y*_t = y_t - x_t'beta (blocks of 64 steps, the waves in turn)
  for (int tb = wave * WAVE; tb < T; tb += 2 * WAVE) {
    const int t = tb + lane;
    double pred = 0.0;
    for (int base = 0; base < p; base += WAVE) {
      const int j = base + lane;
      const double bj = (j < p) ? beta[j] : 0.0;
      unsigned long long mk = __ballot(bj != 0.0);
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const double b = rl(bj, l);
        pred += P.X[(size_t)(base + l) * T + (t < T ? t : T - 1)] * b;
      }
    }
    if (t < T) w0[t] = P.y[t] - pred;
  }

  SSTAMP(1);
  // ---- 2. the normals of simulate_forward, in stream order.  t = 0: the initial
  // state of every state model (rmvn_mt draws every component; the local level
  // model draws rnorm_mt(a0, sd0): nothing if sd0 == 0), then the observation;
  // t >= 1: the state errors (local level: one if sigma != 0; local linear trend:
  // two, always; seasonal: one if sigma != 0; autoregression: one, always --
  // rnorm_mt(rng) * sigma, ArStateModel.cpp:85-90), then the observation.
  const int dH = (sqrtH != 0.0);
  const int d0 = (TREND == 1) ? (Q.S->P0[0] != 0.0 ? 1 : 0) : 2;
  const int nfirst = d0 + S.ns + S.na + dH;
  const int dT = (TREND == 1) ? (sdv[0] != 0.0 ? 1 : 0) : 2;
  const int dS = (SEAS && sdv[2] != 0.0) ? 1 : 0;
  const int dA = AR ? 1 : 0;
  const int nper = dT + dS + dA + dH;
  const int N = nfirst + (T - 1) * nper;
  status = stream_normals(s_lds.norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, P.pos_state[chain], N,
                          szz, &P.pos_state[chain], ss_slot_serve(P));
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }
  __syncthreads();
  SSTAMP(2);

  const bool mylane = lane < m;
  // which variance parameter drives this lane's state error (the seasonal one moves with the cursor)
  const double sig_tr = (lane == 0) ? sig2[0] : ((TREND == 2 && lane == 1) ? sig2[1] : 0.0);
  const double sd_tr = (lane == 0) ? sdv[0] : ((TREND == 2 && lane == 1) ? sdv[1] : 0.0);
  double a0l = 0.0, P0l = 0.0;
#pragma unroll
  for (int i = 0; i < SSM_MAX; ++i) if (lane == i) { a0l = Q.S->a0[i]; P0l = Q.S->P0[i]; }
  double *blk = s_blk[wave];

  // ---- 3. forward, the two waves side by side (neither needs the other's results):
  //   wave 1: the variances P_t -> F_t, K_t (ScalarMarginalDistribution::update, the
  //           part that does not look at the data);
  //   wave 0: simulate alpha+_t, y+_t and w_t = y*_t - y+_t.
  // Time runs in blocks of 64 steps: a block's scalar inputs sit one step per lane
  // (read with v_readlane), its state-sized series in LDS, and what a block
  // produces goes out in one coalesced piece.
  if (wave == 1) {
    // P lives in LDS (s_P[row * PLD + column], PLD = 17: a lane per column and a lane per
    // row are both free of bank conflicts; both indices in the rotating layout): the rows
    // and columns a step touches move with the cursor, which registers cannot follow.
    // A step is P <- T (P - PZ PZ' / F) T' + RQR in TWO phases (the predicted form of
    // round 3 took seven LDS round trips a step, and the pass is bound by exactly those):
    //   column phase: lane k reads ITS column, takes the rank-one term off all of it
    //     ((PZ_i PZ_k) / F: the same product in entry (i, k) and (k, i)) and applies T from the
    //     left -- the trend's row 0 += row 1, the seasonal block's new first row;
    //   row phase: lane k reads the entries of ITS row that T' from the right touches, and
    //     what it writes -- P'(k, 0), P'(k, new first seasonal), P'(k, first lag) -- is
    //     PZ_k of the next step (P is symmetric), which therefore needs no read of its own.
    for (int e = lane; e < SSM_MAX * PLD; e += WAVE) s_P[e] = 0.0;
    wave_lds_sync();
    if (mylane) s_P[lane * (PLD + 1)] = P0l;
    wave_lds_sync();
    int c = 0;
    const int col = mylane ? lane : 0;
    // PZ_k of step 0: P0 is diagonal
    double PZ = (mylane && (lane == 0 || (SEAS && lane == S.s0) || (AR && lane == S.a0))) ? P0l : 0.0;
    if (lane < SSM_MAX) s_pz[lane] = PZ;
    wave_lds_sync();
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const int ob_l = (tt < T && P.observed[tt]) ? 1 : 0;
      double F_l = 1.0;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const int cn = SEAS ? cursor_prev(c, S.ns) : 0;   // the next layout's cursor
        const int rw = S.s0 + cn;                          // row / column of the new first seasonal component
        const double F = zdot<SEAS, AR>(S, PZ, c) + H;
        if (!(F > 0.0)) { status = CHAIN_FORECAST_VARIANCE; break; }
        const double Finv = 1.0 / F;
        const double TPZ = vecT<TREND, SEAS, AR>(S, PZ, lane, c, phl);
        if (mylane) blk[s * m + lane] = obs ? TPZ * Finv : 0.0;
        if (lane == s) F_l = F;
        if (!AR) {
          // -- ONE phase (no autoregression block).  Lane k reads ITS column, takes the
          // rank-one term off, applies T from the left: u = (T P~)[:, k].  T' from the right
          // leaves a column alone unless its own row of T is not a unit row -- so for every
          // lane but the block-first ones (lane 0 of a local linear trend, the lane of the new
          // first seasonal component) u IS the new column, and the entries of the special
          // columns are, P being symmetric, entries of the other lanes' u: lane k writes
          // P'(0, k) also to (k, 0) and P'(rw, k) also to (k, rw).  What is left are the four
          // entries where two special rows meet: sums of the neighbours' u (two v_readlane)
          // and one total over the seasonal lanes.  The next step's PZ_k comes out of the
          // same registers.  One LDS round trip a step instead of two.
          double v[SSM_MAX];
#pragma unroll
          for (int i = 0; i < SSM_MAX; ++i) v[i] = s_P[i * PLD + col];
          if (obs) {
            double pz[SSM_MAX];
#pragma unroll
            for (int i = 0; i < SSM_MAX; ++i) pz[i] = s_pz[i];
#pragma unroll
            for (int i = 0; i < SSM_MAX; ++i) v[i] -= (pz[i] * PZ) * Finv;
          }
          double u0 = (TREND == 2) ? v[0] + v[1] : v[0];
          double cs = 0.0;
          if (SEAS) {
            double t[SSM_MAX];
#pragma unroll
            for (int q = 0; q < SSM_MAX; ++q) t[q] = (TREND + q < SSM_MAX) ? v[(TREND + q) & (SSM_MAX - 1)] : 0.0;
            cs = -((((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                   (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]))));
          }
          if (!mylane) { u0 = 0.0; cs = 0.0; }
          // where the special rows meet (valid in lane 0 / everywhere)
          double p00 = u0 + sig2[0], p0rw = cs;
          if (TREND == 2) {
            p00 = (u0 + rl(u0, 1)) + sig2[0];
            p0rw = cs + rl(cs, 1);
          }
          const double p0rw_u = rl(p0rw, 0);                                               // P'(0, rw) = P'(rw, 0)
          const double prwrw = SEAS ? -row_total(S.seasonal(lane) ? cs : 0.0) + sig2[2] : 0.0;   // P'(rw, rw)
          const bool first_trend = TREND == 2 && lane == 0, first_seas = SEAS && lane == rw;
          if (mylane) {
            if (!first_trend && !first_seas) {
              v[0] = (lane == 0) ? p00 : u0;            // (a local level's lane 0: an ordinary column, + RQR)
              if (TREND == 2 && lane == 1) v[1] += sig2[1];
#pragma unroll
              for (int i = 0; i < SSM_MAX; ++i) s_P[i * PLD + lane] = v[i];
              if (SEAS) {
                s_P[rw * PLD + lane] = cs;
                s_P[lane * PLD + rw] = cs;
              }
              if (TREND == 2) s_P[lane * PLD] = u0;
            } else if (first_trend) {
              s_P[0] = p00;
              if (SEAS) { s_P[rw] = p0rw; s_P[rw * PLD] = p0rw; }
            } else {
              s_P[rw * (PLD + 1)] = prwrw;
            }
          }
          // PZ_k of the next step: P'(0, k) + P'(new first seasonal, k)
          double nz = (lane == 0) ? p00 : u0;
          if (SEAS) nz += first_trend ? p0rw : (first_seas ? prwrw : cs);
          if (SEAS && first_seas) nz = p0rw_u + prwrw;
          PZ = mylane ? nz : 0.0;
          if (lane < SSM_MAX) s_pz[lane] = PZ;
          wave_lds_sync();
          c = cn;
          continue;
        }
        // -- the column phase
        {
          double v[SSM_MAX];
#pragma unroll
          for (int i = 0; i < SSM_MAX; ++i) v[i] = s_P[i * PLD + col];
          if (obs) {
            // (the other lanes' PZ from LDS, all lanes the same addresses: 8 wide reads instead of 32 v_readlane)
            double pz[SSM_MAX];
#pragma unroll
            for (int i = 0; i < SSM_MAX; ++i) pz[i] = s_pz[i];
#pragma unroll
            for (int i = 0; i < SSM_MAX; ++i) v[i] -= (pz[i] * PZ) * Finv;
          }
          if (TREND == 2) v[0] += v[1];
          double cs = 0.0;
          if (SEAS) {
            // (without an autoregression block the rows past the seasonal block are zero: no guard)
            double t[SSM_MAX];
#pragma unroll
            for (int q = 0; q < SSM_MAX; ++q) t[q] = ((!AR || q < S.ns) && TREND + q < SSM_MAX) ? v[(TREND + q) & (SSM_MAX - 1)] : 0.0;
            cs = -((((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                   (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]))));
          }
          if (mylane) {
            if (obs) {
#pragma unroll
              for (int i = 0; i < SSM_MAX; ++i) s_P[i * PLD + lane] = v[i];
            } else if (TREND == 2) {
              s_P[lane] = v[0];
            }
            if (SEAS) s_P[rw * PLD + lane] = cs;
          }
          wave_lds_sync();
          if (AR) {
            // the autoregression block's rows (logical order): from the last lag down, moving each
            // entry one place on as it is read
            if (mylane) {
              if (S.na <= 2) {
                // (one or two lags -- what AddAr is used with -- in a straight line: both entries
                // loaded together; the rolled loop below waits for every entry in turn)
                const bool two = S.na == 2;
                const double x0 = s_P[S.a0 * PLD + lane];
                const double x1 = two ? s_P[(S.a0 + 1) * PLD + lane] : 0.0;
                const double p0 = s_phi[0], p1 = two ? s_phi[1] : 0.0;
                double ca = 0.0;
                if (two) ca += p1 * x1;
                ca += p0 * x0;
                if (two) s_P[(S.a0 + 1) * PLD + lane] = x0;
                s_P[S.a0 * PLD + lane] = ca;
              } else {
                double ca = 0.0;
#pragma nounroll
                for (int q = S.na - 1; q >= 0; --q) {
                  const double x = s_P[(S.a0 + q) * PLD + lane];
                  ca += s_phi[q] * x;
                  if (q + 1 < S.na) s_P[(S.a0 + q + 1) * PLD + lane] = x;
                }
                s_P[S.a0 * PLD + lane] = ca;
              }
            }
            wave_lds_sync();
          }
        }
        // -- the row phase
        {
          double *row = s_P + col * PLD;
          const double w0 = row[0];
          double n0 = w0;
          if (TREND == 2) {
            const double w1 = row[1];
            n0 = w0 + w1;
            if (lane == 1) row[1] = w1 + sig2[1];
          }
          if (lane == 0) n0 += sig2[0];
          double nz = n0;
          if (SEAS) {
            double t[SSM_MAX];
#pragma unroll
            for (int q = 0; q < SSM_MAX; ++q) t[q] = ((!AR || q < S.ns) && TREND + q < SSM_MAX) ? row[(TREND + q) & (SSM_MAX - 1)] : 0.0;
            double cr = -((((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                          (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]))));
            if (lane == rw) cr += sig2[2];
            if (mylane) row[rw] = cr;
            nz += cr;
          }
          if (AR) {
            double ca = 0.0;
            if (mylane) {
              if (S.na <= 2) {
                const bool two = S.na == 2;
                const double x0 = row[S.a0];
                const double x1 = two ? row[S.a0 + 1] : 0.0;
                const double p0 = s_phi[0], p1 = two ? s_phi[1] : 0.0;
                if (two) ca += p1 * x1;
                ca += p0 * x0;
                if (two) row[S.a0 + 1] = x0;
              } else {
#pragma nounroll
                for (int q = S.na - 1; q >= 0; --q) {
                  const double x = row[S.a0 + q];
                  ca += s_phi[q] * x;
                  if (q + 1 < S.na) row[S.a0 + q + 1] = x;
                }
              }
              if (lane == S.a0) ca += sig2a;
              row[S.a0] = ca;
            }
            nz += ca;
          }
          if (mylane && (TREND == 2 || lane == 0)) row[0] = n0;
          PZ = mylane ? nz : 0.0;
          if (lane < SSM_MAX) s_pz[lane] = PZ;
          wave_lds_sync();
        }
        c = cn;
      }
      if (status != CHAIN_OK) break;
      blk_store(gK + (size_t)tb * m, blk, nstep * m, lane);
      if (tt < T) sres[tt] = F_l;
      // the block is out: the filter (wave 0, once it has simulated) follows a block behind
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&s_vprog, tb / WAVE + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (status != CHAIN_OK) {
      if (lane == 0) {
        s_flag = status;
        __hip_atomic_store(&s_vprog, V_FAILED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      return;
    }
    if (scans) return;   // (the last pass is wave 0's alone: time across the lanes)
  } else if (scans) {
    // ---- simulate alpha+_t, y+_t and w_t = y*_t - y+_t with time across the lanes
    double alpha0;
    {
      double z = 0.0;
      if (mylane) {
        if (lane < TREND) z = (lane < d0) ? szz[lane] : 0.0;
        else z = szz[d0 + (lane - TREND)];
      }
      alpha0 = mylane ? sqrt(P0l) * z + a0l : 0.0;   // simulate_initial_state: mean_i + sd_i z_i
    }
    PathCarry PC;
    const double zs1 = (SEAS && dS && T > 1) ? szz[nfirst + dT] : 0.0;
    path_start<TREND, SEAS>(S, lane, alpha0, sdv[2] * zs1, PC);
    const int cidx = SEAS ? WAVE - (S.ns + 1) + lane % (S.ns + 1) : 0;
    // the AR block (one or two lags): a_t = phi_0 a_{t-1} + phi_1 a_{t-2} + sigma z_t
    const double ar_init = AR ? rl(alpha0, S.a0) : 0.0;
    double ar1 = (AR && S.na == 2) ? rl(alpha0, S.a0 + 1) : 0.0, ar2 = 0.0;   // the values at t - 1, t - 2
    const double ph0 = AR ? s_phi[0] : 0.0, ph1 = (AR && S.na == 2) ? s_phi[1] : 0.0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const bool in_l = tt < T;
      const double ys_l = in_l ? w0[tt] : 0.0;
      const int nb_l = (tt == 0) ? 0 : nfirst + (tt - 1) * nper;
      double z0_l = 0.0, z1_l = 0.0, zs_l = 0.0, za_l = 0.0, zh_l = 0.0;
      if (in_l && tt > 0) {
        int o = nb_l;
        if (dT >= 1) z0_l = szz[o++];
        if (dT == 2) z1_l = szz[o++];
        if (dS) zs_l = szz[o++];
        if (dA) za_l = szz[o++];
        if (dH) zh_l = szz[o];
      } else if (in_l) {
        if (dH) zh_l = szz[d0 + S.ns + S.na];
      }
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
      PathStep a;
      path_block<TREND, SEAS>(S, lane, cidx, tt, in_l, sdv[0] * z0_l, sdv[1] * z1_l, sdv[2] * zs_l, PC, s_xw, a);
      double a_ar0 = 0.0, a_ar1 = 0.0;
      if (AR) {
        ar_block(s_aw, lane, tb, nstep, sda * za_l, ar_init, ph0, ph1, ar1, ar2);
        a_ar0 = s_aw[16 + lane];
        a_ar1 = s_aw[15 + lane];
        wave_lds_sync();
      }
      const double yplus = ((a.lev + (SEAS ? a.seas[0] : 0.0)) + (AR ? a_ar0 : 0.0)) + sqrtH * zh_l;   // simulate_adjusted_observation
      if (in_l) {
        w0[tt] = ys_l - yplus;
        blk[lane * m] = a.lev;
        if (TREND == 2) blk[lane * m + 1] = a.slo;
        if (SEAS) {
#pragma unroll
          for (int i = 0; i < SSM_MAX - 1; ++i)
            if (i < S.ns) blk[lane * m + TREND + i] = a.seas[i];
        }
        if (AR) {
          blk[lane * m + S.a0] = a_ar0;
          if (S.na == 2) blk[lane * m + S.a0 + 1] = a_ar1;
        }
      }
      blk_store(gst + (size_t)tb * m, blk, nstep * m, lane);   // (alpha+ in logical order: only the last pass reads it)
    }
  } else {
    double alpha = 0.0;
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const bool in_l = tt < T;
      const double ys_l = in_l ? w0[tt] : 0.0;
      const int nb_l = (tt == 0) ? 0 : nfirst + (tt - 1) * nper;
      double z0_l = 0.0, z1_l = 0.0, zs_l = 0.0, za_l = 0.0, zh_l = 0.0;
      if (in_l && tt > 0) {
        int o = nb_l;
        if (dT >= 1) z0_l = szz[o++];
        if (dT == 2) z1_l = szz[o++];
        if (dS) zs_l = szz[o++];
        if (dA) za_l = szz[o++];
        if (dH) zh_l = szz[o];
      } else if (in_l) {
        if (dH) zh_l = szz[d0 + S.ns + S.na];
      }
      double w_l = 0.0;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        if (tb + s == 0) {
          // simulate_initial_state: mean_i + sd_i z_i
          double z = 0.0;
          if (mylane) {
            // (the blocks follow one another: seasonal and autoregression lanes alike)
            if (lane < TREND) z = (lane < d0) ? szz[lane] : 0.0;
            else z = szz[d0 + (lane - TREND)];
          }
          alpha = mylane ? sqrt(P0l) * z + a0l : 0.0;
        } else {
          // simulate_next_state: T alpha + eta
          const double z0 = rl(z0_l, s);
          const int cn = SEAS ? cursor_prev(c, S.ns) : 0;
          alpha = vecT<TREND, SEAS, AR>(S, alpha, lane, c, phl);
          if (TREND == 2) alpha += sd_tr * ((lane == 0) ? z0 : rl(z1_l, s));
          else alpha += sd_tr * z0;
          if (SEAS) { if (lane == S.s0 + cn) alpha += sdv[2] * rl(zs_l, s); }
          if (AR) { if (lane == S.a0) alpha += rl(za_l, s) * sda; }
          c = cn;
        }
        const double yplus = zdot<SEAS, AR>(S, alpha, c) + sqrtH * rl(zh_l, s);   // simulate_adjusted_observation
        const double w = rl(ys_l, s) - yplus;
        if (lane == s) w_l = w;
        if (mylane) blk[s * m + lane] = alpha;
      }
      blk_store(gst + (size_t)tb * m, blk, nstep * m, lane);
      if (in_l) w0[tt] = w_l;
    }
  }
  SSTAMP(3);
  SSTAMP(4);

  double r = 0.0;
  if (wave == 0) {
  // ---- 3b. the filter on w = y* - y+ (the data filter minus the simulation
  // filter; they share the gains): v - v+ = w - Z'(a - a+); a - a+ <- T (a - a+) + K (v - v+)
  {
    double delta = 0.0;
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const bool in_l = tt < T;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
      // (the gains and F_t of this block: wave 1 is somewhere ahead, or about to be)
      int vp;
      while ((vp = __hip_atomic_load(&s_vprog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) <= tb / WAVE)
        __builtin_amdgcn_s_sleep(8);
      if (vp == V_FAILED) {
        if (lane == 0) P.status[chain] = __hip_atomic_load(&s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
      }
      blk_load(blk, gK + (size_t)tb * m, nstep * m, lane);
      const double w_l = in_l ? w0[tt] : 0.0, F_l = in_l ? sres[tt] : 1.0;
      const int ob_l = (in_l && P.observed[tt]) ? 1 : 0;
      double ef_l = 0.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double K = mylane ? blk[s * m + lane] : 0.0;
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const double e = obs ? rl(w_l, s) - zdot<SEAS, AR>(S, delta, c) : 0.0;
        if (lane == s) ef_l = obs ? e / F_l : 0.0;
        delta = vecT<TREND, SEAS, AR>(S, delta, lane, c, phl) + K * e;
        if (SEAS) c = cursor_prev(c, S.ns);
      }
      __builtin_amdgcn_wave_barrier();
      if (in_l) w0[tt] = ef_l;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  SSTAMP(5);
  // ---- 4. backward: fast_disturbance_smooth for d = r - r+:
  // r_{t-1} = T' r_t + Z ((v_t - v+_t) / F_t - K_t' r_t), r_{T-1} = 0.  r_t is in the
  // layout of step t + 1.
  for (int tb = ((T - 1) / WAVE) * WAVE; tb >= 0; tb -= WAVE) {
    const int tt = tb + lane;
    const bool in_l = tt < T;
    const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
    blk_load(blk, gK + (size_t)tb * m, nstep * m, lane);
    const double ef_l = in_l ? w0[tt] : 0.0;
    double d0_l = 0.0, d1_l = 0.0, d2_l = 0.0, d3_l = 0.0;
    int c1 = SEAS ? cursor_at(tb + nstep, S.ns) : 0;   // layout of r at the block's last step
#pragma nounroll
    for (int s = nstep - 1; s >= 0; --s) {
      const double K = mylane ? blk[s * m + lane] : 0.0;
      const int c0 = SEAS ? (c1 + 1 == S.ns ? 0 : c1 + 1) : 0;   // c_t from c_{t+1}
      // r_t at the rows that carry state error: what the correction pass needs
      const double q0 = rl(r, 0);
      if (lane == s) d0_l = q0;
      if (TREND == 2) { const double q1 = rl(r, 1); if (lane == s) d1_l = q1; }
      if (SEAS) { const double q2 = rl(r, S.s0 + c1); if (lane == s) d2_l = q2; }
      if (AR) { const double q3 = rl(r, S.a0); if (lane == s) d3_l = q3; }
      const double kr = row_total(K * r);
      const double coef = rl(ef_l, s) - kr;
      r = vecTt<TREND, SEAS, AR>(S, r, lane, c1, phl);
      if (lane == 0 || (SEAS && lane == S.s0 + c0) || (AR && lane == S.a0)) r += coef;
      if (!mylane) r = 0.0;
      c1 = c0;
    }
    __builtin_amdgcn_wave_barrier();
    if (in_l) {
      gd[tt] = d0_l;
      if (TREND == 2) gd[(size_t)T + tt] = d1_l;
      if (SEAS) gd[(size_t)2 * T + tt] = d2_l;
      if (AR) gd[(size_t)3 * T + tt] = d3_l;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  }
  SSTAMP(6);
  if (scans) {
    // ---- 5'. the mean correction, the state draw and every sufficient statistic with
    // time across the lanes (wave 0; wave 1 has left)
    const double mc0 = mylane ? P0l * r : 0.0;   // a0 + P0 r0 - (a0 + P0 r0+), time 0's layout
    PathCarry PC;
    path_start<TREND, SEAS>(S, lane, mc0, (SEAS && T > 1) ? sig2[2] * gd[(size_t)2 * T] : 0.0, PC);
    const int cidx = SEAS ? WAVE - (S.ns + 1) + lane % (S.ns + 1) : 0;
    double c_lev = 0.0, c_slo = 0.0, c_sum = 0.0;   // the state of the previous block's last step
    double ss0 = 0.0, ss1 = 0.0, ss2 = 0.0, yty = 0.0, nobs = 0.0;
    const double ar_init = AR ? rl(mc0, S.a0) : 0.0;
    double ar1 = (AR && S.na == 2) ? rl(mc0, S.a0 + 1) : 0.0, ar2 = 0.0;
    const double ph0 = AR ? s_phi[0] : 0.0, ph1 = (AR && S.na == 2) ? s_phi[1] : 0.0;
    double c_a0 = 0.0, c_a1 = 0.0;
    double x00 = 0.0, x01 = 0.0, x11 = 0.0, xy0 = 0.0, xy1 = 0.0, ayy = 0.0;   // the ArModel's sufficient statistics
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const bool in_l = tt < T;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
      blk_load(blk, gst + (size_t)tb * m, nstep * m, lane);
      const bool dd = in_l && tt > 0;
      const double d0_l = dd ? gd[tt - 1] : 0.0;
      const double d1_l = (dd && TREND == 2) ? gd[(size_t)T + tt - 1] : 0.0;
      const double d2_l = (dd && SEAS) ? gd[(size_t)2 * T + tt - 1] : 0.0;
      const double d3_l = (dd && AR) ? gd[(size_t)3 * T + tt - 1] : 0.0;
      const double y_l = in_l ? P.y[tt] : 0.0;
      const bool ob_l = in_l && P.observed[tt];
      PathStep c;
      path_block<TREND, SEAS>(S, lane, cidx, tt, in_l, sig2[0] * d0_l, sig2[1] * d1_l, sig2[2] * d2_l, PC, s_xw, c);
      double sa0 = 0.0, sa1 = 0.0;
      if (AR) {
        ar_block(s_aw, lane, tb, nstep, sig2a * d3_l, ar_init, ph0, ph1, ar1, ar2);
        if (in_l) {
          sa0 = blk[lane * m + S.a0] + s_aw[16 + lane];
          if (S.na == 2) sa1 = blk[lane * m + S.a0 + 1] + s_aw[15 + lane];
        }
        wave_lds_sync();
      }
      double st0 = 0.0, st1 = 0.0, sea0 = 0.0, bsum = 0.0;
      double sea[SSM_MAX - 1];
      if (in_l) {
        st0 = blk[lane * m] + c.lev;
        if (TREND == 2) st1 = blk[lane * m + 1] + c.slo;
      }
      if (SEAS) {
#pragma unroll
        for (int i = 0; i < SSM_MAX - 1; ++i) {
          sea[i] = (in_l && i < S.ns) ? blk[lane * m + TREND + i] + c.seas[i] : 0.0;
          bsum += sea[i];
        }
        sea0 = sea[0];
      }
      // the state before this step: the lane below (the previous block's last lane for lane 0)
      double p0 = __shfl_up(st0, 1), p1 = __shfl_up(st1, 1), pb = __shfl_up(bsum, 1);
      if (lane == 0) { p0 = c_lev; p1 = c_slo; pb = c_sum; }
      if (dd) {
        if (TREND == 1) {
          const double diff = st0 - p0;
          ss0 += diff * diff;
        } else {
          // err = now - T then (MvnSuf of the errors: what is published is sum err^2)
          const double e0 = st0 - (p0 + p1), e1 = st1 - p1;
          ss0 += e0 * e0;
          ss1 += e1 * e1;
        }
        if (SEAS) {
          const double dl = sea0 - (-1.0 * pb);   // now[0] + sum(then) over the block
          ss2 += dl * dl;
        }
      }
      if (AR) {
        // add_mixture_data(now[0], then, 1.0): xtx += then then', xty += now[0] then, yty += now[0]^2
        double pa0 = __shfl_up(sa0, 1), pa1 = __shfl_up(sa1, 1);
        if (lane == 0) { pa0 = c_a0; pa1 = c_a1; }
        if (dd) {
          x00 += pa0 * pa0; x01 += pa0 * pa1; x11 += pa1 * pa1;
          xy0 += sa0 * pa0; xy1 += sa0 * pa1;
          ayy += sa0 * sa0;
        }
        c_a0 = rl(sa0, WAVE - 1); c_a1 = rl(sa1, WAVE - 1);
      }
      const double resid = ob_l ? y_l - ((st0 + (SEAS ? sea0 : 0.0)) + (AR ? sa0 : 0.0)) : 0.0;
      if (ob_l) { yty += resid * resid; nobs += 1.0; }
      wave_lds_sync();
      if (in_l) {
        blk[lane * m] = st0;
        if (TREND == 2) blk[lane * m + 1] = st1;
        if (SEAS) {
#pragma unroll
          for (int i = 0; i < SSM_MAX - 1; ++i)
            if (i < S.ns) blk[lane * m + TREND + i] = sea[i];
        }
        if (AR) {
          blk[lane * m + S.a0] = sa0;
          if (S.na == 2) blk[lane * m + S.a0 + 1] = sa1;
        }
        sres[tt] = resid;
      }
      blk_store(gst + (size_t)tb * m, blk, nstep * m, lane);
      c_lev = rl(st0, WAVE - 1); c_slo = rl(st1, WAVE - 1); c_sum = rl(bsum, WAVE - 1);
    }
    ss0 = wave_total(ss0); ss1 = wave_total(ss1); ss2 = wave_total(ss2);
    yty = wave_total(yty); nobs = wave_total(nobs);
    if (AR) {
      x00 = wave_total(x00); x01 = wave_total(x01); x11 = wave_total(x11);
      xy0 = wave_total(xy0); xy1 = wave_total(xy1); ayy = wave_total(ayy);
    }
    SSTAMP(7);
#ifdef BA_KSTAMPS
    if (chain == 0 && lane == 0 && draw_variances)
      printf("ssm phases (cycles, wave 0): variances %lld ystar %lld normals %lld sim %lld wait-for-P %lld filter %lld backward %lld correction %lld\n",
             kph[0], kph[1], kph[2], kph[3], kph[4], kph[5], kph[6], kph[7]);
#endif
    if (lane == 0) {
      Q.M.var_n[Q.at(chain, 0)] = (double)(T - 1);
      Q.M.var_ss[Q.at(chain, 0)] = ss0;
      if (TREND == 2) {
        Q.M.var_n[Q.at(chain, 1)] = (double)(T - 1);
        Q.M.var_ss[Q.at(chain, 1)] = ss1;
      }
      if (SEAS) {
        Q.M.var_n[Q.at(chain, 2)] = (double)(T - 1);
        Q.M.var_ss[Q.at(chain, 2)] = ss2;
      }
      if (AR) {
        double *suf = Q.ar_suf(chain);
        suf[0] = x00;
        suf[AR_SUF_XTY] = xy0;
        if (S.na == 2) {
          suf[1] = x01; suf[SSM_MAX] = x01; suf[SSM_MAX + 1] = x11;
          suf[AR_SUF_XTY + 1] = xy1;
        }
        suf[AR_SUF_YTY] = ayy;
        suf[AR_SUF_N] = (double)(T - 1);
      }
      P.yty[chain] = yty;
      P.nobs[chain] = nobs;
      P.status[chain] = status;
    }
    return;
  }
  // ---- 5. forward: the mean correction E(alpha | y) - E(alpha | y+), the state
  // draw, the state models' and the regression's sufficient statistics -- by BOTH waves:
  // wave 0 runs the recursion of the correction and turns a block of alpha+ (LDS) into the
  // block of state draws; wave 1 (its variance pass long over) follows one block behind with
  // everything that only READS the draws: the state models' sufficient statistics, the
  // residuals, the copy in logical order that goes out.  Two block buffers take turns.
  if (wave == 0) {
    double mc = P0l * r;          // a0 + P0 r0 - (a0 + P0 r0+)
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane, b = tb / WAVE;
      const bool in_l = tt < T;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
      double *buf = s_blk[b & 1];
      while (__hip_atomic_load(&s_cdone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b - 1)
        __builtin_amdgcn_s_sleep(4);
      blk_load(buf, gst + (size_t)tb * m, nstep * m, lane);
      const bool dd = in_l && tt > 0;
      const double d0_l = dd ? gd[tt - 1] : 0.0;
      const double d1_l = (dd && TREND == 2) ? gd[(size_t)T + tt - 1] : 0.0;
      const double d2_l = (dd && SEAS) ? gd[(size_t)2 * T + tt - 1] : 0.0;
      const double d3_l = (dd && AR) ? gd[(size_t)3 * T + tt - 1] : 0.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double ap = mylane ? buf[s * m + lane] : 0.0;
        if (tb + s > 0) {
          const int cn = SEAS ? cursor_prev(c, S.ns) : 0;
          mc = vecT<TREND, SEAS, AR>(S, mc, lane, c, phl);
          if (TREND == 2) mc += sig_tr * ((lane == 0) ? rl(d0_l, s) : rl(d1_l, s));
          else mc += sig_tr * rl(d0_l, s);
          if (SEAS) { if (lane == S.s0 + cn) mc += sig2[2] * rl(d2_l, s); }
          if (AR) { if (lane == S.a0) mc += sig2a * rl(d3_l, s); }
          c = cn;
        }
        if (mylane) buf[s * m + lane] = ap + mc;
      }
      wave_lds_sync();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&s_cprog, b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    SSTAMP(7);
#ifdef BA_KSTAMPS
    if (chain == 0 && lane == 0 && draw_variances)
      printf("ssm phases (cycles, wave 0): variances %lld ystar %lld normals %lld sim %lld wait-for-P %lld filter %lld backward %lld correction %lld\n",
             kph[0], kph[1], kph[2], kph[3], kph[4], kph[5], kph[6], kph[7]);
#endif
    return;
  }
  double prev = 0.0;            // state_{t-1} (its own layout)
  double suf0 = 0.0, suf2 = 0.0;
  double mv_ybar = 0.0, mv_sumsq = 0.0, mv_n = 0.0;   // MvnSuf of the trend errors (lanes 0, 1)
  double yty = 0.0, nobs = 0.0;
  // ArModel's NeRegSuf of now[a0] on then[a0 ..]: lane a0 + i keeps xty_i and row i of xtx,
  // the row in LDS (s_P is free by now), at s_axx[i * PLD + q]
  double axy = 0.0, ayy = 0.0;
  double *s_axx = s_P;
  if (AR) {
    for (int e2 = lane; e2 < SSM_MAX * PLD; e2 += WAVE) s_axx[e2] = 0.0;
    wave_lds_sync();
  }
  {
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane, b = tb / WAVE;
      const bool in_l = tt < T;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
      double *buf = s_blk[b & 1];
      const double y_l = in_l ? P.y[tt] : 0.0;
      const int ob_l = (in_l && P.observed[tt]) ? 1 : 0;
      // (wave 0 only leaves early when THIS wave's variance pass failed, and then this wave has left too)
      while (__hip_atomic_load(&s_cprog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= b)
        __builtin_amdgcn_s_sleep(8);
      double res_l = 0.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double st = mylane ? buf[s * m + lane] : 0.0;
        if (tb + s > 0 && SEAS) c = cursor_prev(c, S.ns);
        if (tb + s > 0) {
          if (TREND == 1) {
            const double diff = st - prev;                 // (lane 0)
            if (lane == 0) suf0 += diff * diff;
          } else {
            // err = now - T then; MvnSuf::update_raw (MvnBase.cpp:71-86), diagonal only
            const double then1 = rl(prev, 1);
            const double err = st - ((lane == 0) ? prev + then1 : prev);
            mv_n += 1.0;
            const double wv = (err - mv_ybar) / mv_n;
            mv_ybar += wv;
            mv_sumsq += wv * wv * (mv_n - 1);
            const double w2 = err - mv_ybar;
            mv_sumsq += w2 * w2;
          }
          if (SEAS) {
            // delta = now[0] + sum(then) over the seasonal block
            const double tot = row_total(S.seasonal(lane) ? prev : 0.0);
            const double dl = st - (-1.0 * tot);
            if (lane == S.s0 + c) suf2 += dl * dl;
          }
          if (AR) {
            // add_mixture_data(now[0], then, 1.0): xtx += then then', xty += now[0] then, yty += now[0]^2
            const double yy = rl(st, S.a0);
            {
              double *row = s_axx + (S.ar(lane) ? lane - S.a0 : 0) * PLD;
#pragma nounroll
              for (int q = 0; q < S.na; ++q) {
                const double pq = rl(prev, S.a0 + q);
                if (S.ar(lane)) row[q] += prev * pq * 1.0;
              }
            }
            axy += (yy * 1.0) * prev;
            ayy += yy * yy * 1.0;
          }
        }
        prev = st;
        if (SEAS) {
          // the state draw goes out in logical order (in place: every lane has read its entry)
          if (mylane && S.seasonal(lane)) {
            const int q = lane - S.s0;
            buf[s * m + S.s0 + (q >= c ? q - c : q - c + S.ns)] = st;
          }
        }
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const double resid = obs ? rl(y_l, s) - zdot<SEAS, AR>(S, st, c) : 0.0;
        if (lane == s) res_l = resid;
        if (obs) { yty += resid * resid; nobs += 1.0; }
      }
      blk_store(gst + (size_t)tb * m, buf, nstep * m, lane);
      if (in_l) sres[tt] = res_l;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&s_cdone, b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  // publish the sufficient statistics
  if (TREND == 2) {
    // center_sumsq(mu = 0)(i, i) = sumsq_ii + n ybar_i^2
    const double ssv = mv_sumsq + mv_ybar * mv_ybar * mv_n;
    if (lane < 2) {
      Q.M.var_n[(size_t)chain * SSG_MAX_VAR + lane] = mv_n;   // (level, slope: variance parameters 0, 1)
      Q.M.var_ss[(size_t)chain * SSG_MAX_VAR + lane] = ssv;
    }
  } else if (lane == 0) {
    Q.M.var_n[Q.at(chain, 0)] = (double)(T - 1);
    Q.M.var_ss[Q.at(chain, 0)] = suf0;
  }
  if (SEAS) {
    // (the lane that accumulated moved with the cursor: sum over the block)
    const double tot = row_total(S.seasonal(lane) ? suf2 : 0.0);
    if (lane == 0) {
      Q.M.var_n[Q.at(chain, 2)] = (double)(T - 1);
      Q.M.var_ss[Q.at(chain, 2)] = tot;
    }
  }
  if (AR) {
    double *suf = Q.ar_suf(chain);
    if (S.ar(lane)) {
      const int i = lane - S.a0;
      for (int q = 0; q < S.na; ++q) suf[i * SSM_MAX + q] = s_axx[i * PLD + q];
      suf[AR_SUF_XTY + i] = axy;
    }
    if (lane == 0) {
      suf[AR_SUF_YTY] = ayy;
      suf[AR_SUF_N] = (double)(T - 1);
    }
  }
  if (lane == 0) {
    P.yty[chain] = yty;
    P.nobs[chain] = nobs;
    P.status[chain] = status;
  }
}

// the template kernel for the launch's shape (ssm_kernel.hip runs the X'e GEMM behind it)
hipError_t launch_ssm_template(hipStream_t stream, const SsParams &P, int draw_variances) {
  const dim3 grid(P.chain_count), block(2 * WAVE);
  const bool seas = P.ssm.tpl_nseasons > 0, ar = P.ssm.tpl_ar_lags > 0;
#define SSM_LAUNCH(TR, SE, AR) hipLaunchKernelGGL((ssm_simsmooth_kernel<TR, SE, AR>), grid, block, 0, stream, P, draw_variances)
#define SSM_LAUNCH_AR(TR, SE) do { if (ar) SSM_LAUNCH(TR, SE, true); else SSM_LAUNCH(TR, SE, false); } while (0)
  if (!seas) {
    if (P.ssm.tpl_trend == 1) SSM_LAUNCH_AR(1, false); else SSM_LAUNCH_AR(2, false);
  } else {
    if (P.ssm.tpl_trend == 1) SSM_LAUNCH_AR(1, true); else SSM_LAUNCH_AR(2, true);
  }
#undef SSM_LAUNCH_AR
#undef SSM_LAUNCH
  return hipGetLastError();
}

}  // namespace boom_amd
