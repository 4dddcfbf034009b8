// Device helpers of the general structural state-space kernel (ssm_kernel.hip: one chain per
// workgroup, any state dimension <= 64): cross-lane moves, the ArPosteriorSampler, the block list and the
// transition's action on lane-distributed vectors.  See ssm_kernel.hip for the model.
#pragma once
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "kalman_params.h"
#include "stream_normals.h"

namespace boom_amd {

namespace {

constexpr int WAVE = 64;

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
// sum over the first 16 lanes (the others hold 0), everywhere
__device__ __forceinline__ double row_total(double x) {
  x += sdpp<0x111, 0xf>(x, 0.0);  // row_shr:1
  x += sdpp<0x112, 0xf>(x, 0.0);
  x += sdpp<0x114, 0xf>(x, 0.0);
  x += sdpp<0x118, 0xf>(x, 0.0);
  return rl(x, 15);
}
// sum over the wave (lanes that do not take part hold 0), everywhere.  SMALL: the state
// has at most 16 components -- one row of lanes
template <bool SMALL>
__device__ __forceinline__ double wsum(double x) {
  x += sdpp<0x111, 0xf>(x, 0.0);
  x += sdpp<0x112, 0xf>(x, 0.0);
  x += sdpp<0x114, 0xf>(x, 0.0);
  x += sdpp<0x118, 0xf>(x, 0.0);
  if (SMALL) return rl(x, 15);
  return ((rl(x, 15) + rl(x, 31)) + rl(x, 47)) + rl(x, 63);
}
// the value of lane - 1 (lane 0: 0) / of lane + 1 (lane 63: 0): wave_shr:1 / wave_shl:1
__device__ __forceinline__ double from_below(double x) { return sdpp<0x138, 0xf>(x, 0.0); }
__device__ __forceinline__ double from_above(double x) { return sdpp<0x130, 0xf>(x, 0.0); }

__device__ __forceinline__ int sprev(int c, int ns) { return c == 0 ? ns - 1 : c - 1; }   // the cursor after a move
__device__ __forceinline__ int snext(int c, int ns) { return c + 1 == ns ? 0 : c + 1; }   // ... before it
// how many of the times 1 .. t start a new season (u % duration == phase)
__device__ __forceinline__ int seasons_started(int t, int duration, int phase) {
  if (t < 0) return 0;
  return (t >= phase ? (t - phase) / duration + 1 : 0) - (phase == 0 ? 1 : 0);
}

// a block of `n` doubles between HBM and LDS, by one wave
__device__ __forceinline__ void blk_load(double *lds, const double *g, int n, int lane) {
  for (int i = lane; i < n; i += WAVE) lds[i] = g[i];
  wave_lds_sync();
}
__device__ __forceinline__ void blk_store(double *g, const double *lds, int n, int lane) {
  wave_lds_sync();
  for (int i = lane; i < n; i += WAVE) g[i] = lds[i];
  wave_lds_sync();
}

// ---- ArPosteriorSampler::draw for one chain, by one (whole) wave.  Vectors sit one
// component per lane (lane i < L), the L x L matrices in LDS at leading dimension
// AR_MAX; every lane reads the same random numbers.
struct ArLds {
  double X[AR_MAX * AR_MAX];    // xtx
  double Lc[AR_MAX * AR_MAX];   // chol(xtx)
  double Lp[AR_MAX * AR_MAX];   // chol(xtx / sigsq)
};
// lower Cholesky factor of `scale` * A (A symmetric, full storage); false: not positive definite
__device__ __forceinline__ bool ar_chol(const double *A, double scale, double *Lc, int L, int lane) {
  for (int j = 0; j < L; ++j) {
    double sacc = 0.0;
    if (lane >= j && lane < L) {
      sacc = A[lane * AR_MAX + j] * scale;
      for (int k = 0; k < j; ++k) sacc -= Lc[lane * AR_MAX + k] * Lc[j * AR_MAX + k];
    }
    const double djj = rl(sacc, j);
    if (!(djj > 0.0)) return false;
    const double d = sqrt(djj);
    if (lane == j) Lc[j * AR_MAX + j] = d;
    else if (lane > j && lane < L) Lc[lane * AR_MAX + j] = sacc / d;
    __builtin_amdgcn_wave_barrier();
  }
  return true;
}
// x: Lc x = b
__device__ __forceinline__ double ar_lsolve(const double *Lc, double b, int L, int lane) {
  double x = 0.0;
  for (int i = 0; i < L; ++i) {
    const double tot = row_total((lane < i) ? Lc[i * AR_MAX + lane] * x : 0.0);
    const double xi = (rl(b, i) - tot) / Lc[i * AR_MAX + i];
    if (lane == i) x = xi;
  }
  return x;
}
// x: Lc' x = b
__device__ __forceinline__ double ar_ltsolve(const double *Lc, double b, int L, int lane) {
  double x = 0.0;
  for (int i = L - 1; i >= 0; --i) {
    const double tot = row_total((lane > i && lane < L) ? Lc[lane * AR_MAX + i] * x : 0.0);
    const double xi = (rl(b, i) - tot) / Lc[i * AR_MAX + i];
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
// draw_sigma.  suf: the block's sufficient statistics; phi_l: the lane's coefficient
// (in: current, out: drawn); *sigsq likewise.
__device__ __forceinline__ int ar_draw(ArLds &W, const double *suf, int L, double prior_df, double prior_ss,
                                       double sigma_max, SeqRng &rng, double &phi_l, double &sigsq, int lane) {
  for (int e = lane; e < AR_MAX * AR_MAX; e += WAVE) W.X[e] = suf[e];
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
      const double ivar = W.X[i * AR_MAX + i];
      const double dot = row_total((lane < L) ? ph * W.X[lane * AR_MAX + i] : 0.0);
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
    if (lane < L) row += W.X[lane * AR_MAX + j] * pj;
  }
  const double quad = row_total((lane < L) ? phi_l * row : 0.0);
  const double lin = row_total((lane < L) ? phi_l * xty : 0.0);
  const double ss = quad - 2 * lin + yty;
  int bad = 0;
  sigsq = d_draw_variance(rng, n + prior_df, ss + prior_ss, sigma_max, &bad);
  return bad ? CHAIN_RNG_BRANCH : CHAIN_OK;
}

// ---- NonzeroMeanAr1Sampler::draw (NonzeroMeanAr1Sampler.cpp:51-155) for a semilocal trend's slope
// model, on its Ar1Suf (NonzeroMeanAr1Model.cpp:39-128; suf = sumsq, sum, cross, n, first, last):
// draw_mu, draw_phi (a normal; truncated to [lo, 1] under force_stationary), draw_sigma -- one
// generator, every lane of the (whole) wave the same numbers.  In / out: mu, phi, sigsq.
__device__ __forceinline__ int semilocal_slope_draw(const double *suf, const double *prior, bool truncate, bool positive,
                                                    double prior_df, double prior_ss, double sigma_max, SeqRng &rng,
                                                    double &mu, double &phi, double &sigsq) {
  const double a1_sumsq = suf[0], a1_sum = suf[1], a1_cross = suf[2], n = suf[3], a1_first = suf[4], a1_last = suf[5];
  const double lag_sumsq = a1_sumsq - a1_last * a1_last;
  const double lag_sum = a1_sum - a1_last;
  const double sum_excluding_first = a1_sum - a1_first;
  const double sumsq_excluding_first = a1_sumsq - a1_first * a1_first;
  {  // draw_mu
    const double psig = prior[1] * prior[1];
    const double omp = 1 - phi;
    double ivar = (1 + (n - 1) * (omp * omp)) / sigsq;
    ivar += 1.0 / psig;
    double mean = (1 - phi) * (sum_excluding_first - phi * lag_sum) + a1_first;
    mean /= sigsq;
    mean += prior[0] / psig;
    mean /= ivar;
    mu = d_rnorm(rng, mean, sqrt(1.0 / ivar));
  }
  {  // draw_phi
    const double psig = prior[3] * prior[3];
    double ivar = lag_sumsq - 2 * lag_sum * mu + (n - 1) * mu * mu;   // centered_lag_sumsq(mu)
    ivar /= sigsq;
    ivar += 1.0 / psig;
    double mean = a1_cross - mu * (sum_excluding_first + lag_sum) + (n - 1) * mu * mu;   // centered_cross(mu)
    mean /= sigsq;
    mean += prior[2] / psig;
    mean /= ivar;
    const double sd = sqrt(1.0 / ivar);
    if (truncate) {
      int bad = 0;
      phi = ar_rtrun_norm_2(rng, mean, sd, positive ? 0.0 : -1.0, 1.0, &bad);
      if (bad) return CHAIN_RNG_BRANCH;
    } else {
      phi = d_rnorm(rng, mean, sd);
    }
  }
  {  // draw_sigma: model_sumsq(mu, phi), df = n
    const double d0 = a1_first - mu, mp = mu * (1 - phi);
    double ss = d0 * d0;
    ss += sumsq_excluding_first - 2 * phi * a1_cross - 2 * (1 - phi) * mu * sum_excluding_first +
          phi * phi * lag_sumsq + 2 * phi * (1 - phi) * mu * lag_sum + (n - 1) * (mp * mp);
    int bad = 0;
    sigsq = d_draw_variance(rng, n + prior_df, ss + prior_ss, sigma_max, &bad);
    if (bad) return CHAIN_RNG_BRANCH;
  }
  return CHAIN_OK;
}

// ---- the block list: lane b of each of three registers holds block b (so that all blocks
// advance in one vector operation and a loop over blocks is a ROLLED loop that fetches its
// block with v_readlane -- eight unrolled copies of every per-block code path made a kernel
// of 60 000 instructions that ran out of the instruction cache).
struct Blocks {
  unsigned desc;   // kind (3 bits) | first (7) | dim (7) | index of its first variance parameter (5) | autoregression slot (3)
  unsigned dp;     // seasonal: duration (16 bits) | phase (16)
  unsigned rc;     // seasonal: (index of the step's transition + 1 - phase) mod duration (16 bits: 0 = it moves) |
                   //           the rotating layout's cursor (16): physical slot of the block's first component
  int nb;
  unsigned always;     // bit b: block b's transition is never the identity (trend, autoregression)
  unsigned seasmask;   // bit b: block b is seasonal
  unsigned armask;     // bit b: block b is an autoregression
  unsigned trigmask;   // bit b: block b is a trig model (pairs of components that rotate)
  unsigned slmask;     // bit b: block b is a semilocal linear trend (level, slope, the slope's long-run mean)
  static __device__ __forceinline__ int kind_of(unsigned d) { return (int)(d & 7u); }
  static __device__ __forceinline__ int first_of(unsigned d) { return (int)((d >> 3) & 127u); }
  static __device__ __forceinline__ int dim_of(unsigned d) { return (int)((d >> 10) & 127u); }
  static __device__ __forceinline__ int var0_of(unsigned d) { return (int)((d >> 17) & 31u); }
  static __device__ __forceinline__ int arx_of(unsigned d) { return (int)((d >> 22) & 7u); }
  // block b's words, wave-uniform
  __device__ __forceinline__ unsigned udesc(int b) const { return (unsigned)__builtin_amdgcn_readlane((int)desc, b); }
  __device__ __forceinline__ unsigned urc(int b) const { return (unsigned)__builtin_amdgcn_readlane((int)rc, b); }
  __device__ __forceinline__ void load(const SsgSpec &Q, int nblocks, int lane) {
    nb = nblocks;
    desc = 0; dp = 1; rc = 0;
    if (lane < nblocks) {
      const SsgBlock &K = Q.blk[lane];
      desc = (unsigned)K.kind | ((unsigned)K.first << 3) | ((unsigned)K.dim << 10) |
             ((unsigned)K.var0 << 17) | ((unsigned)(K.ar_index < 0 ? 0 : K.ar_index) << 22);
      dp = (unsigned)K.duration | ((unsigned)K.phase << 16);
    }
    const int kd = kind_of(desc);
    always = (unsigned)__ballot(kd == SSG_LOCAL_LINEAR_TREND || kd == SSG_AR || kd == SSG_TRIG || kd == SSG_SEMILOCAL);
    slmask = (unsigned)__ballot(kd == SSG_SEMILOCAL);
    seasmask = (unsigned)__ballot(kd == SSG_SEASONAL);
    armask = (unsigned)__ballot(kd == SSG_AR);
    trigmask = (unsigned)__ballot(kd == SSG_TRIG);
  }
  // the layout of time t; the transitions are walked from index t + shift on (0: the
  // transition OUT of the time, T_t; -1: the one INTO it, T_{t-1})
  __device__ __forceinline__ void seek(int t, int shift) {
    if (kind_of(desc) == SSG_SEASONAL) {
      const int d = (int)(dp & 0xffffu), ph = (int)(dp >> 16), n = dim_of(desc);
      const int q = seasons_started(t, d, ph) % n;
      const int c = q == 0 ? 0 : n - q;
      int r = (t + shift + 1 - ph) % d;
      if (r < 0) r += d;
      rc = (unsigned)r | ((unsigned)c << 16);
    }
  }
  // bit b = block b's transition of this step is not the identity
  __device__ __forceinline__ unsigned moving() const {
    return always | (unsigned)__ballot(kind_of(desc) == SSG_SEASONAL && (rc & 0xffffu) == 0u);
  }
  // one step on / back: the layout after (before) the transitions `mv`, the next (previous) transition's phase
  __device__ __forceinline__ void advance(unsigned mv, int lane) {
    if (kind_of(desc) == SSG_SEASONAL) {
      int r = (int)(rc & 0xffffu) + 1, c = (int)(rc >> 16);
      if (r == (int)(dp & 0xffffu)) r = 0;
      if ((mv >> lane) & 1u) c = sprev(c, dim_of(desc));
      rc = (unsigned)r | ((unsigned)c << 16);
    }
  }
  __device__ __forceinline__ void retreat(unsigned mv, int lane) {
    if (kind_of(desc) == SSG_SEASONAL) {
      int r = (int)(rc & 0xffffu), c = (int)(rc >> 16);
      r = r == 0 ? (int)(dp & 0xffffu) - 1 : r - 1;
      if ((mv >> lane) & 1u) c = snext(c, dim_of(desc));
      rc = (unsigned)r | ((unsigned)c << 16);
    }
  }
};

// per-lane constants of the lane's component
struct LaneInfo {
  int blk, kind, first, dim;    // its block (kind 0: the lane holds no component)
  int cur;                      // seasonal: its block's cursor (a per-lane copy)
  double phi;                   // autoregression: the lane's coefficient; semilocal trend: the slope's AR(1) coefficient (every lane of the block)
  double tc = 0.0, ts = 0.0;    // trig: cosine and sine of the lane's pair
  __device__ __forceinline__ bool moves(unsigned mv) const { return kind == SSG_SEASONAL && ((mv >> blk) & 1u); }
  // trig: is this the second component of its pair?
  __device__ __forceinline__ bool todd(int lane) const { return ((lane - first) & 1) != 0; }
  // is this a lane Z selects in its block (the block's first component; trig: every pair's first)?
  template <bool GLOB = true>
  __device__ __forceinline__ bool zsel(int lane) const {
    if (GLOB && kind == SSG_TRIG) return !todd(lane);
    return kind != 0 && lane == first + cur;
  }
};

// Z'x
template <bool SMALL, bool GLOB = true>
__device__ __forceinline__ double zdot(const LaneInfo &L, double x, int lane) {
  return wsum<SMALL>(L.template zsel<GLOB>(lane) ? x : 0.0);
}
// y = T x for a vector held one component per lane, in the layout the cursors say; mv:
// bit b = block b's transition moves at this step (seasonal: the step into a new season);
// the result is in the next layout (the caller advances the cursors)
// GLOB: the list holds a trig or a semilocal-linear-trend block (round 6); lists without them run
// the instance that does not carry their code in its per-step loops
template <bool SMALL, bool GLOB = false>
__device__ __forceinline__ double vecT(const Blocks &B, const LaneInfo &L, double x, int lane, unsigned mv) {
  double y = x;
  // (cross-lane moves read INACTIVE lanes as nothing: they stay outside the per-lane branches)
  const double above = from_above(x);
  if (L.kind == SSG_LOCAL_LINEAR_TREND && lane == L.first) y = x + above;
  // autoregression: new[0] = phi'old, new[i] = old[i - 1]  (AutoRegressionTransitionMatrix, SparseMatrix.cpp:1261-1310)
  if (B.armask) {
    const double below = from_below(x);
    if (L.kind == SSG_AR) y = below;
    unsigned am = B.armask;
    while (am) {
      const int b = __ffs((int)am) - 1;
      am &= am - 1;
      const double tot = wsum<SMALL>(L.blk == b ? L.phi * x : 0.0);
      if (L.blk == b && lane == L.first) y = tot;
    }
  }
  // semilocal trend: (level, slope, mean) -> (level + slope, phi slope + (1 - phi) mean, mean)
  // (SemilocalLinearTrendMatrix::multiply, SemilocalLinearTrend.cpp:36-46)
  if (GLOB && L.kind == SSG_SEMILOCAL) {
    if (lane == L.first) y = x + above;
    else if (lane == L.first + 1) y = L.phi * x + (1 - L.phi) * above;
  }
  // trig: every pair rotates, (x0, x1) -> (c x0 + s x1, -s x0 + c x1)  (the DenseMatrix blocks of
  // TrigStateModel.cpp:144-153)
  if (GLOB && B.trigmask) {
    const double below = from_below(x);
    if (L.kind == SSG_TRIG) y = L.todd(lane) ? -L.ts * below + L.tc * x : L.tc * x + L.ts * above;
  }
  // seasonal, a step into a new season: the slot of the component that drops out receives
  // -(sum over the block)
  unsigned sm = mv & B.seasmask;
  while (sm) {
    const int b = __ffs((int)sm) - 1;
    sm &= sm - 1;
    const double tot = wsum<SMALL>(L.blk == b ? x : 0.0);
    if (L.blk == b && lane == L.first + sprev(L.cur, L.dim)) y = -tot;
  }
  return y;
}
// y = T' x; the cursors are those of x's layout (time t + 1); mv as above for the step t -> t + 1
template <bool GLOB = false>
__device__ __forceinline__ double vecTt(const Blocks &B, const LaneInfo &L, double x, int lane, unsigned mv) {
  double y = x;
  const double below = from_below(x);
  if (L.kind == SSG_LOCAL_LINEAR_TREND && lane == L.first + 1) y = below + x;
  if (B.armask) {
    // out[i] = phi_i x[0] + x[i + 1]  (Tmult, SparseMatrix.cpp:1286-1295)
    const double above = from_above(x);
    unsigned am = B.armask;
    while (am) {
      const int b = __ffs((int)am) - 1;
      am &= am - 1;
      const double firstv = rl(x, Blocks::first_of(B.udesc(b)));
      if (L.blk == b) y = L.phi * firstv + ((lane + 1 < L.first + L.dim) ? above : 0.0);
    }
  }
  // semilocal trend: Tmult (SemilocalLinearTrend.cpp:65-76): (r0, r1, r2) -> (r0, r0 + phi r1, (1 - phi) r1 + r2)
  if (GLOB && L.kind == SSG_SEMILOCAL) {
    if (lane == L.first + 1) y = below + L.phi * x;
    else if (lane == L.first + 2) y = (1 - L.phi) * below + x;
  }
  if (GLOB && B.trigmask) {
    // the rotations' transposes: (x0, x1) -> (c x0 - s x1, s x0 + c x1)
    const double above = from_above(x);
    if (L.kind == SSG_TRIG) y = L.todd(lane) ? L.ts * below + L.tc * x : L.tc * x + -L.ts * above;
  }
  unsigned sm = mv & B.seasmask;
  while (sm) {
    const int b = __ffs((int)sm) - 1;
    sm &= sm - 1;
    const int c1 = Blocks::first_of(B.udesc(b)) + (int)(B.urc(b) >> 16);
    const double firstv = rl(x, c1);
    if (L.blk == b) y = (lane == c1) ? -firstv : x - firstv;
  }
  return y;
}
__device__ __forceinline__ void advance(Blocks &B, LaneInfo &L, unsigned mv, int lane) {
  B.advance(mv, lane);
  if (L.moves(mv)) L.cur = sprev(L.cur, L.dim);
}
__device__ __forceinline__ void retreat(Blocks &B, LaneInfo &L, unsigned mv, int lane) {
  B.retreat(mv, lane);
  if (L.moves(mv)) L.cur = snext(L.cur, L.dim);
}
// the layout of time t (see Blocks::seek)
__device__ __forceinline__ void seek(Blocks &B, LaneInfo &L, int t, int shift) {
  B.seek(t, shift);
  // (every lane takes part: a lane that is switched off is read as 0 by the others)
  const unsigned mine = (unsigned)__shfl((int)B.rc, L.blk < 0 ? 0 : L.blk);
  __builtin_amdgcn_wave_barrier();
  L.cur = (L.kind == SSG_SEASONAL) ? (int)(mine >> 16) : 0;
}

}  // namespace

}  // namespace boom_amd
