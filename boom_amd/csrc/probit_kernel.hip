// BinomialProbitSpikeSlabSampler's data augmentation for many chains (SURVEY 8f
// row f3, the probit member):
//   BinomialProbitSpikeSlabSampler::impute_latent_data
//       (Models/Glm/PosteriorSamplers/BinomialProbitSpikeSlabSampler.cpp:55-69)
//   BinomialProbitDataImputer::impute          (BinomialProbitDataImputer.cpp:30-73)
//   rtrun_norm_mt / trun_norm_mt / TnSampler   (distributions/trun_norm.cpp:36-47,
//                                               :108-241)
// One thread per (chain, observation): eta = x_i'beta over the chain's included
// variables, the sum of the observation's latent normals -- each trial a
// standard normal truncated to the side of zero its outcome says, or the
// central-limit draw when there are more than clt_threshold of a kind -- goes
// to z[chain][i]; X'z for all chains is then ONE MFMA GEMM (the same one the
// state-space path uses for X'e), and the inclusion / coefficient draws are the
// SpikeSlabSampler mode of the sweep kernels on (X'NX, X'z).
//
// The reference reads the sampler's RNG in sequence, observation after
// observation, a data-dependent number of uniforms each.  Here observation i of
// sweep s reads the chain's imputer stream (id 8) from position (s n + i) * 4096:
// a counter-based stream makes the observations independent.  (The oracle's
// Philox mode does the same; its MT mode, pinned on the reference, reads in
// sequence.)
#include <hip/hip_runtime.h>

#include "ktimer.h"

#include "device_rng.h"
#include "probit_params.h"

namespace boom_amd {

namespace {

// how many uniforms a slot of an imputer's substream hands out before the draw goes on in the
// spill stream (device_rng.h): the whole stride, or what ba_set_slot_limit asked for (tests)
__device__ __forceinline__ uint32_t slot_serve(const ProbitParams &P, uint32_t stride) {
  return (P.slot_limit > 0 && (uint32_t)P.slot_limit < stride) ? (uint32_t)P.slot_limit : stride;
}
// TnSampler (Samplers/TnSampler.cpp): bounded adaptive rejection for a standard normal given
// x > a, a > 0 (logf = -x^2 / 2).  The hull's points are x_0 = a < x_1 < ...; after every
// rejected candidate the reference recomputes all knots and all segment integrals
// (add_point / update_cdf) and probes the cdf and the knots with std::lower_bound: restated
// here as written, on the same inputs.
//
// The hull lives in LDS, laid out [array][point][slot] (a lane's bank depends on its slot
// only, whatever point it touches): per-thread arrays indexed by numbers that differ from lane
// to lane were 2.5 KB of scratch memory per lane and 64 cache lines per access.  Two sizes:
// TN_SLOTS hulls of TN_LDS_CAP points for the draws of the workgroup's SECOND phase (see the
// kernel), TN_BIG_SLOTS of TN_BIG_CAP points for the few that outgrow those -- an observation
// that does is done again from its first draw with a large hull (the same numbers: same
// expressions on the same inputs, the stream is positional); a hull of more than TN_BIG_CAP
// points is reported (CHAIN_RNG_BRANCH), as it always was.
enum : int { TN_LDS_CAP = 16, TN_SLOTS = 64, TN_BIG_CAP = 64, TN_BIG_SLOTS = 4 };
struct TnHull {
  double *base;   // this thread's slot of the hull arrays: array a, point i at base[(a * cap + i) * stride]
  int stride;     // slots side by side
  int cap;        // points
};
// returns false when the hull outgrew its capacity (the caller redoes the observation)
__device__ __forceinline__ bool tn_draw_lds(SeqRng &rng, double a, const TnHull &H, double *out) {
  auto yv = [](double x) { return __dmul_rn(__dmul_rn(-.5, x), x); };   // (a stored value in the reference: rounded)
  const int st = H.stride;
  double *xs = H.base, *kn = xs + H.cap * st, *cdf = kn + H.cap * st;
#define AT(arr, k) arr[(k) * st]
  const double y0 = yv(a);
  int n = 1;
  AT(xs, 0) = a; AT(kn, 0) = a;
  for (int level = 0; level <= 1001; ++level) {
    {  // update_cdf
      double last = 0;
      for (int k = 0; k < n; ++k) {
        const double z = AT(xs, k), d = -z, y = yv(z) - y0, dinv = 1.0 / d;
        const double inc1 = (k == n - 1) ? 0 : dinv * exp(y - d * z + d * AT(kn, k + 1));
        const double inc2 = dinv * exp(y - d * z + d * AT(kn, k));
        last = last + inc1 - inc2;
        AT(cdf, k) = last;
      }
    }
    const double u = d_runif(rng, 0.0, AT(cdf, n - 1));
    int k = 0;
    {  // std::lower_bound(cdf, cdf + n, u)
      int count = n;
      while (count > 0) {
        const int step = count / 2;
        if (AT(cdf, k + step) < u) { k += step + 1; count -= step + 1; }
        else count = step;
      }
    }
    const double xk = AT(xs, k);
    double cand;
    if (k + 1 == n) cand = AT(kn, n - 1) + d_rexp(rng, -1 * -AT(xs, n - 1));
    else cand = d_rtrun_exp(rng, -1 * -xk, AT(kn, k), AT(kn, k + 1));
    const double target = yv(cand);
    const double hull = yv(xk) + -xk * (cand - xk);
    const double logu = hull - d_rexp(rng, 1.0);
    if (logu < target) { *out = cand; return true; }
    if (n >= H.cap) return false;
    int pos = 0;
    {  // std::lower_bound(knots, knots + n, cand)
      int count = n;
      while (count > 0) {
        const int step = count / 2;
        if (AT(kn, pos + step) < cand) { pos += step + 1; count -= step + 1; }
        else count = step;
      }
    }
    if (pos == 0) return false;   // (a candidate at the truncation point itself: never)
    for (int i = n; i > pos; --i) AT(xs, i) = AT(xs, i - 1);
    for (int i = n; i > pos + 1; --i) AT(kn, i) = AT(kn, i - 1);
    AT(xs, pos) = cand;
    ++n;
    for (int i = pos; i <= pos + 1 && i < n; ++i) {
      const double xl = AT(xs, i - 1), xr = AT(xs, i);
      double ans = (yv(xl) - -xl * xl) - (yv(xr) - -xr * xr);
      ans /= (-xr - -xl);
      AT(kn, i) = ans;
    }
  }
#undef AT
  return false;
}

// trun_norm_mt(rng, a): a standard normal given x > a.  ARS false: the caller knows a <= 0
// (the kernel's first phase); *overflow: the hull outgrew its slot.
template <bool ARS>
__device__ __forceinline__ double trun_norm_std(SeqRng &rng, double a, const TnHull &H, bool *overflow) {
  if (!ARS || a <= 0) {
    for (;;) {
      const double x = d_norm_rand(rng);
      if (x > a) return x;
    }
  }
  double z = a;
  if (!tn_draw_lds(rng, a, H, &z)) *overflow = true;
  return z;
}
template <bool ARS>
__device__ __forceinline__ double rtrun_norm(SeqRng &rng, double mu, double a, bool gt, const TnHull &H,
                                             bool *overflow) {
  return gt ? mu + trun_norm_std<ARS>(rng, a - mu, H, overflow)
            : mu - trun_norm_std<ARS>(rng, mu - a, H, overflow);   // (sigma = 1)
}
__device__ __forceinline__ double log_pnorm_std(double x, bool lower) {
  const double z = lower ? -x : x;
  return log(0.5 * erfc(z / 1.4142135623730951));
}
// trun_norm_moments(mu, 1, 0, positive_support, ...)
__device__ __forceinline__ void trun_norm_moments(double mu, bool positive, double *mean, double *variance) {
  const double log_phi_const = -0.918938533204672741780329736406;
  const double t = 0.0 - mu;
  if (positive) {
    const double phi_ratio = exp((log_phi_const - .5 * t * t) - log_pnorm_std(t, false));
    *mean = mu + phi_ratio;
    *variance = 1 - phi_ratio * (phi_ratio - t);
  } else {
    const double phi_ratio = exp((log_phi_const - .5 * t * t) - log_pnorm_std(t, true));
    *mean = mu - phi_ratio;
    *variance = 1 - t * phi_ratio - phi_ratio * phi_ratio;
  }
  if (*variance < 0) *variance = 0;
}

// trun_norm_moments(mu, sigma, 0, positive_support, ...) (distributions/trun_norm.cpp:243-269)
__device__ __forceinline__ void trun_norm_moments_s(double mu, double sigma, bool positive, double *mean,
                                                    double *variance) {
  const double log_phi_const = -0.918938533204672741780329736406;
  const double sigsq = sigma * sigma;
  const double t = (0.0 - mu) / sigma;
  if (positive) {
    const double phi_ratio = exp((log_phi_const - .5 * t * t) - log_pnorm_std(t, false));
    *mean = mu + sigma * phi_ratio;
    const double delta = phi_ratio * (phi_ratio - t);
    *variance = sigsq * (1 - delta);
  } else {
    const double phi_ratio = exp((log_phi_const - .5 * t * t) - log_pnorm_std(t, true));
    *mean = mu - sigma * phi_ratio;
    *variance = sigsq * (1 - t * phi_ratio - phi_ratio * phi_ratio);
  }
  if (*variance < 0) *variance = 0;
}

// BOOM::binomial_distribution(n, p)(rng) (distributions/BinomialDistribution.cpp:24-158;
// what Rmath::rbinom_mt calls, Bmath/rbinom.cpp:66-69): Kachitvichyanukul and
// Schmeiser's BTPE for n p >= 30, sequential inversion below -- the statements and the
// uniforms they consume in the reference's order.
__device__ __noinline__ unsigned d_rbinom(SeqRng &rng, unsigned n, double pp) {
  double c = 0, fm = 0, npq = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
  double xl = 0, xll = 0, xlr = 0, xm = 0, xr = 0;
  int m = 0, ix = 0;
  const double p = (pp < 1. - pp) ? pp : 1. - pp;
  const double q = 1. - p;
  const double np = n * p;
  const double r = p / q;
  const double g = r * (n + 1);
  bool done = false;
  if (np < 30) {
    const double qn = pow(q, (double)n);
    while (!done) {
      ix = 0;
      double f = qn;
      double u = rng();
      for (;;) {
        if (u < f) { done = true; break; }
        if (ix > 110) break;
        u -= f;
        ix++;
        f *= (g / ix - r);
      }
    }
  } else {
    const double ffm = np + p;
    m = (int)ffm;
    fm = m;
    npq = np * q;
    p1 = (int)(2.195 * sqrt(npq) - 4.6 * q) + 0.5;
    xm = fm + 0.5;
    xl = xm - p1;
    xr = xm + p1;
    c = 0.134 + 20.5 / (15.3 + fm);
    double al = (ffm - xl) / (ffm - xl * p);
    xll = al * (1.0 + 0.5 * al);
    al = (xr - ffm) / (xr * q);
    xlr = al * (1.0 + 0.5 * al);
    p2 = p1 * (1.0 + c + c);
    p3 = p2 + c / xll;
    p4 = p3 + c / xlr;
  }
  while (!done) {
    const double u = rng() * p4;
    double v = rng();
    if (u <= p1) {  // triangular region
      ix = (int)(xm - p1 * v + u);
      break;
    }
    if (u <= p2) {  // parallelogram region
      const double x = xl + (u - p1) / c;
      v = v * c + 1.0 - fabs(xm - x) / p1;
      if (v > 1.0 || v <= 0.) continue;
      ix = (int)x;
    } else if (u > p3) {  // right tail
      ix = (int)(xr - log(v) / xlr);
      if ((unsigned)ix > n) continue;
      v = v * (u - p3) * xlr;
    } else {  // left tail
      ix = (int)(xl + log(v) / xll);
      if (ix < 0) continue;
      v = v * (u - p2) * xll;
    }
    const int k = abs(ix - m);
    if (k <= 20 || k >= npq / 2 - 1) {
      double f = 1.0;
      if (m < ix) {
        for (int i = m + 1; i <= ix; i++) f *= (g / i - r);
      } else if (m != ix) {
        for (int i = ix + 1; i <= m; i++) f /= (g / i - r);
      }
      if (v <= f) break;
    } else {
      const double amaxp = (k / npq) * ((k * (k / 3. + 0.625) + 0.1666666666666) / npq + 0.5);
      const double ynorm = -1.0 * k * k / (2.0 * npq);
      const double alv = log(v);
      if (alv < ynorm - amaxp) break;
      if (alv <= ynorm + amaxp) {
        const double x1 = ix + 1, f1 = fm + 1.0, z = n + 1 - fm, w = n - ix + 1.0;
        const double z2 = z * z, x2 = x1 * x1, f2 = f1 * f1, w2 = w * w;
        if (alv <= xm * log(f1 / x1) + (n - m + 0.5) * log(z / w) + (ix - m) * log(w * p / (x1 * q)) +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / f2) / f2) / f2) / f2) / f1 / 166320.0 +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / z2) / z2) / z2) / z2) / z / 166320.0 +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / x2) / x2) / x2) / x2) / x1 / 166320.0 +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / w2) / w2) / w2) / w2) / w / 166320.)
          break;
      }
    }
  }
  if (pp > 0.5) ix = (int)n - ix;
  return (unsigned)ix;
}

// Rmath::rmultinom_mt (Bmath/rmultinom.cpp:82-136) for probabilities that sum to one
__device__ __forceinline__ void d_rmultinom9(SeqRng &rng, int n, const double (&prob)[9], int (&rN)[9]) {
  double p_tot = 0.;
#pragma unroll
  for (int k = 0; k < 9; ++k) { p_tot += prob[k]; rN[k] = 0; }
  if (n == 0) return;
  for (int k = 0; k < 8; ++k) {
    const double pp = prob[k] / p_tot;
    rN[k] = (int)d_rbinom(rng, (unsigned)n, pp);
    n -= rN[k];
    if (n <= 0) return;
    p_tot -= prob[k];
  }
  rN[8] = n;
}

// The chain's included variables and their coefficients, in ascending order (the
// order x_i'beta is summed in), to LDS; returns how many there are (beyond
// PROBIT_KMAX only counted).  All 256 threads: 256 variables per round, a
// variable's place = included ones in earlier rounds + earlier waves + earlier lanes.
__device__ __forceinline__ int included_coefficients(const ProbitParams &P, int chain, int *s_idx,
                                                     double *s_beta) {
  __shared__ int s_wave_count[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint8_t *g = P.gamma + (size_t)chain * P.p;
  const double *b = P.beta + (size_t)chain * P.p;
  int base = 0;
  for (int j0 = 0; j0 < P.p; j0 += 256) {
    const int j = j0 + tid;
    const bool inc = j < P.p && g[j] != 0;
    const unsigned long long m = __ballot(inc);
    if (lane == 0) s_wave_count[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = s_wave_count[w];
      before += (w < wave) ? c : 0;
      total += c;
    }
    if (inc) {
      const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < PROBIT_KMAX) { s_idx[pos] = j; s_beta[pos] = b[j]; }
    }
    base += total;
    __syncthreads();
  }
  return base;
}

}  // namespace

// One observation's latent sum (BinomialProbitDataImputer::impute, .cpp:30-73).  ARS false:
// the caller has checked that no draw of this observation is on the far side of its mean.
template <bool ARS>
__device__ __forceinline__ double probit_impute_one(const ProbitParams &P, int chain, int i, double eta, long nt,
                                                    long y, const TnHull &H, bool *overflow, bool *bad) {
  SeqRng rng = SeqRng::slot(PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 8u},
                            P.sweep * (uint64_t)P.n + (uint64_t)i, PROBIT_STRIDE, slot_serve(P, PROBIT_STRIDE));
  double mean, variance, ans = 0.0;
  if (y > P.clt_threshold) {
    trun_norm_moments(eta, true, &mean, &variance);
    ans += d_rnorm(rng, y * mean, sqrt(y * variance));
  } else {
    for (long t = 0; t < y && !*overflow; ++t) ans += rtrun_norm<ARS>(rng, eta, 0.0, true, H, overflow);
  }
  if (nt - y > P.clt_threshold) {
    trun_norm_moments(eta, false, &mean, &variance);
    ans += d_rnorm(rng, (nt - y) * mean, sqrt((nt - y) * variance));
  } else {
    for (long t = 0; t < nt - y && !*overflow; ++t) ans += rtrun_norm<ARS>(rng, eta, 0.0, false, H, overflow);
  }
  // (a draw that outruns its slot of the stream goes on in the slot's spill stream, device_rng.h)
  if (rng.overran()) *bad = true;
  return ans;
}

// grid = (ceil(n / 256), chains), block = 256.
//
// Phases.  A truncated normal on the near side of its mean (the cut at or below it) is a
// rejection loop over plain normals; on the far side -- an outcome the linear predictor
// speaks against -- it is the adaptive-rejection sampler: a hull, exponentials, logarithms,
// several candidates.  With one thread per observation every wavefront ran both loops to the
// depth of its unluckiest lane.  So: (1) every thread computes its linear predictor and, for
// a one-trial observation on the near side, tries the FIRST candidate by Kinderman-Ramage's
// first branch -- straight-line code, two thirds of all observations end there --; what is
// left is queued: (1b) near-side observations, done from their first draw by the first three
// wavefronts, 64 to a chunk; (2) far-side observations by the fourth wavefront, each lane a
// hull of 16 points in LDS; (3) what outgrew 16 points -- rare -- again, four at a time,
// with hulls of 64 points.  An observation's draws read its own slot of the chain's stream
// whichever thread makes them, from its first position: the same numbers as before.
__global__ __launch_bounds__(256) void probit_impute_kernel(ProbitParams P) {
  const int chain = (int)blockIdx.y, i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  // Other workgroups of this chain may write the status word during this launch
  // (CHAIN_RNG_BRANCH, CHAIN_MODEL_TOO_LARGE below): the decision to skip a chain in
  // error is taken ONCE per workgroup and shared, so that every wave of the workgroup
  // reaches the barriers below or none does.
  __shared__ int s_status;
  if (threadIdx.x == 0) s_status = __atomic_load_n(P.status + chain, __ATOMIC_RELAXED);
  __syncthreads();
  if (s_status != CHAIN_OK) return;
  // the chain's included variables, once per workgroup -- and, once every thread has its
  // linear predictor, the same memory is the hulls
  constexpr size_t COEF_BYTES = PROBIT_KMAX * (sizeof(int) + sizeof(double));
  constexpr size_t HULL_BYTES = 3 * sizeof(double) * (TN_LDS_CAP * TN_SLOTS + TN_BIG_CAP * TN_BIG_SLOTS);
  __shared__ __align__(16) unsigned char s_mem[COEF_BYTES > HULL_BYTES ? COEF_BYTES : HULL_BYTES];
  __shared__ double s_eta[256];
  __shared__ unsigned char s_queue[256], s_near[256], s_again[256];
  __shared__ int s_nqueue, s_nnear, s_nagain;
  double *s_beta = reinterpret_cast<double *>(s_mem);
  int *s_idx = reinterpret_cast<int *>(s_mem + PROBIT_KMAX * sizeof(double));
  double *s_hull = reinterpret_cast<double *>(s_mem);
  double *s_big = s_hull + 3 * TN_LDS_CAP * TN_SLOTS;
  if (threadIdx.x == 0) { s_nqueue = 0; s_nnear = 0; s_nagain = 0; }
  const int k = included_coefficients(P, chain, s_idx, s_beta);
  if (k > PROBIT_KMAX) {
    if (threadIdx.x == 0 && blockIdx.x == 0) P.status[chain] = CHAIN_MODEL_TOO_LARGE;
    return;
  }
  const TnHull none{nullptr, 0, 0};
  bool bad = false;
  // ---- (1)
  if (i < P.n) {
    double eta = 0.0;
    for (int m = 0; m < k; ++m) eta += P.X[(size_t)s_idx[m] * P.n + i] * s_beta[m];
    const long nt = lround(P.ntrials[i]), y = lround(P.y[i]);
    // a success is drawn above 0 around eta (the far side if eta < 0), a failure below
    const bool far = (y > 0 && y <= P.clt_threshold && eta < 0.0) ||
                     (nt - y > 0 && nt - y <= P.clt_threshold && eta > 0.0);
    bool resolved = false;
    if (!far && nt == 1 && P.clt_threshold >= 1) {
      // one trial, near side: the first candidate by Kinderman-Ramage's first branch (two
      // uniforms, 88 % of normals) -- straight-line code for the whole wavefront.  The
      // other branches' loops and the later candidates are what made every wavefront wait
      // for its unluckiest lane: an observation that meets one goes to the queue and is done
      // from its first draw there (the stream is positional: the same numbers)
      SeqRng rng = SeqRng::slot(PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 8u},
                                P.sweep * (uint64_t)P.n + (uint64_t)i, PROBIT_STRIDE, slot_serve(P, PROBIT_STRIDE));
      const bool gt = y == 1;
      const double cut = gt ? 0.0 - eta : eta - 0.0;   // (rtrun_norm: a - mu / mu - a)
      // (ONE candidate: a second and third straight-line try settle another tenth of the
      // observations and measured no gain -- 5.96 vs 5.89 ms per round)
      for (int c = 0; c < 1 && slot_serve(P, PROBIT_STRIDE) >= 2; ++c) {
        const double u1 = rng();
        if (!(u1 < 0.884070402298758)) break;
        const double u2 = rng();
        const double x = 2.216035867166471 * (1.131131635444180 * u1 + u2 - 1);   // (d_norm_rand's first return)
        if (x > cut) {
          double ans = 0.0;
          ans += gt ? eta + x : eta - x;
          P.z[(size_t)chain * P.n + i] = ans;
          resolved = true;
          break;
        }
      }
    }
    if (!resolved) {
      s_eta[threadIdx.x] = eta;
      if (far) s_queue[atomicAdd(&s_nqueue, 1)] = (unsigned char)threadIdx.x;
      else     s_near[atomicAdd(&s_nnear, 1)] = (unsigned char)threadIdx.x;
    }
  }
  __syncthreads();   // (the coefficients are no longer needed: their memory becomes the hulls)
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  const int first = (int)(blockIdx.x * blockDim.x);
  // ---- (1b) the near-side observations the first candidate did not settle (and those of
  // several trials), dense: chunks of 64 over the first three wavefronts
  if (wave < 3) {
    const int nnear = s_nnear;
    for (int base = 64 * wave; base < nnear; base += 192) {
      const int t = base + lane;
      if (t < nnear) {
        const int li = s_near[t], ii = first + li;
        bool overflow = false;
        P.z[(size_t)chain * P.n + ii] = probit_impute_one<false>(P, chain, ii, s_eta[li], lround(P.ntrials[ii]),
                                                                 lround(P.y[ii]), none, &overflow, &bad);
      }
    }
    if (bad) P.status[chain] = CHAIN_RNG_BRANCH;
    return;
  }
  // ---- (2) the far-side observations, by the last wavefront
  const int nqueue = s_nqueue;
  for (int base = 0; base < nqueue; base += TN_SLOTS) {
    const int t = base + lane;
    if (t < nqueue) {
      const int li = s_queue[t], ii = first + li;
      const TnHull H{s_hull + lane, TN_SLOTS, TN_LDS_CAP};
      bool overflow = false;
      const double ans = probit_impute_one<true>(P, chain, ii, s_eta[li], lround(P.ntrials[ii]), lround(P.y[ii]), H,
                                                 &overflow, &bad);
      if (overflow) s_again[atomicAdd(&s_nagain, 1)] = (unsigned char)li;
      else P.z[(size_t)chain * P.n + ii] = ans;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // ---- (3)
  const int nagain = __hip_atomic_load(&s_nagain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (int base = 0; base < nagain; base += TN_BIG_SLOTS) {
    const int t = base + lane;
    if (lane < TN_BIG_SLOTS && t < nagain) {
      const int li = s_again[t], ii = first + li;
      const TnHull H{s_big + lane, TN_BIG_SLOTS, TN_BIG_CAP};
      bool overflow = false;
      const double ans = probit_impute_one<true>(P, chain, ii, s_eta[li], lround(P.ntrials[ii]), lround(P.y[ii]), H,
                                                 &overflow, &bad);
      if (overflow) bad = true;   // (a hull of more than 64 points: reported, never mishandled)
      P.z[(size_t)chain * P.n + ii] = ans;
    }
  }
  if (bad) P.status[chain] = CHAIN_RNG_BRANCH;
}

// BinomialLogitAuxmixSampler's imputation (BinomialLogitAuxmixSampler.cpp:77-97,
// BinomialLogitCltDataImputer::impute_small_sample, BinomialLogitDataImputer.cpp:
// 128-144): per trial a logistic draw truncated to the side of zero its outcome
// says (inverse cdf, one uniform: distributions/trun_logit.cpp:163-174), the
// component of the nine-normal mixture that approximates the logistic density
// (NormalMixtureApproximation.cpp:280-290, :416-424; one uniform), and from it the
// trial's precision.  z[chain][i] = sum of latent * precision, w[chain][i] = sum
// of precisions.  Exactly two uniforms per trial; an observation with more than
// clt_threshold trials takes the large-sample branch below (16 binomial draws and a
// normal one, a data-dependent number of uniforms).  Observation i of sweep s reads
// the chain's worker stream (id 9) from position (s n + i) * LOGIT_STRIDE.
__global__ __launch_bounds__(256) void logit_impute_kernel(ProbitParams P) {
  const double MIX_SIGMA[9] = {0.88437229872213, 1.16097607474416, 1.28021991084306,
                               1.3592552924727,  1.67589879794907, 2.20287232043947,
                               2.20507148325819, 2.91944313615144, 3.90807611741308};
  const double MIX_WEIGHT[9] = {0.038483985581272, 0.13389889791451,  0.0657842076622429,
                                0.105680086433879, 0.345939491553619, 0.0442261124345564,
                                0.193289780660134, 0.068173066865908, 0.00452437089387876};
  const int chain = (int)blockIdx.y, i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  __shared__ int s_status;   // (one decision per workgroup: see probit_impute_kernel)
  if (threadIdx.x == 0) s_status = __atomic_load_n(P.status + chain, __ATOMIC_RELAXED);
  __syncthreads();
  if (s_status != CHAIN_OK) return;
  __shared__ int s_idx[PROBIT_KMAX];
  __shared__ double s_beta[PROBIT_KMAX];
  const int k = included_coefficients(P, chain, s_idx, s_beta);
  if (k > PROBIT_KMAX) {
    if (threadIdx.x == 0 && blockIdx.x == 0) P.status[chain] = CHAIN_MODEL_TOO_LARGE;
    return;
  }
  if (i >= P.n) return;
  double eta = 0.0;
  for (int m = 0; m < k; ++m) eta += P.X[(size_t)s_idx[m] * P.n + i] * s_beta[m];
  const long nt = lround(P.ntrials[i]), ys = lround(P.y[i]);
  SeqRng rng = SeqRng::slot(PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 9u},
                            P.sweep * (uint64_t)P.n + (uint64_t)i, LOGIT_STRIDE, slot_serve(P, LOGIT_STRIDE));
  double sum = 0.0, info = 0.0;
  if (nt > P.clt_threshold) {
    // BinomialLogitCltDataImputer::impute_large_sample (BinomialLogitDataImputer.cpp:
    // 155-211): how many failures / successes belong to each mixture component (two
    // multinomial draws), then one normal draw for the information-weighted sum from the
    // truncated-normal moments of the occupied cells
    double p0[9], p1[9];
    int N0[9], N1[9];
    const double xz = (0 - eta) / 1.0;
    const double neg_support = 1 / (1 + exp(-xz)), pos_support = 1 / (1 + exp(xz));   // plogis(0, eta, 1, lower / upper)
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      const double z = (0 - eta) / MIX_SIGMA[m];
      p0[m] = MIX_WEIGHT[m] / neg_support * (0.5 * erfc(-z / 1.4142135623730951));
      p1[m] = MIX_WEIGHT[m] / pos_support * (0.5 * erfc(z / 1.4142135623730951));
    }
#pragma unroll
    for (int m = 0; m < 9; ++m) { s0 += p0[m]; s1 += p1[m]; }
#pragma unroll
    for (int m = 0; m < 9; ++m) { p0[m] /= s0; p1[m] /= s1; }
    d_rmultinom9(rng, (int)(nt - ys), p0, N0);
    d_rmultinom9(rng, (int)ys, p1, N1);
    double simulation_mean = 0.0, simulation_variance = 0.0;
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      const int total_obs = N0[m] + N1[m];
      if (total_obs == 0) continue;
      const double sigsq = MIX_SIGMA[m] * MIX_SIGMA[m], sig4 = sigsq * sigsq;
      info += total_obs / sigsq;
      double tmean, tvar;
      if (N0[m] > 0) {
        trun_norm_moments_s(eta, MIX_SIGMA[m], false, &tmean, &tvar);
        simulation_mean += N0[m] * tmean / sigsq;
        simulation_variance += N0[m] * tvar / sig4;
      }
      if (N1[m] > 0) {
        trun_norm_moments_s(eta, MIX_SIGMA[m], true, &tmean, &tvar);
        simulation_mean += N1[m] * tmean / sigsq;
        simulation_variance += N1[m] * tvar / sig4;
      }
    }
    sum = d_rnorm(rng, simulation_mean, sqrt(simulation_variance));
  } else {
    const double cutpoint_prob = 1 / (1 + exp(-(0 - eta)));
    for (long t = 0; t < nt; ++t) {
      const bool success = t < ys;
      const double u = success ? d_runif(rng, cutpoint_prob, 1.0) : d_runif(rng, 0.0, cutpoint_prob);
      const double latent = (0.0 + 1.0 * log(u / (1. - u))) + eta;
      const double v = latent - eta;
      double wsp[9], mx = -__builtin_huge_val(), nc = 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const double xs = (v - 0.0) / MIX_SIGMA[c];
        wsp[c] = log(MIX_WEIGHT[c]) + -(0.918938533204672741780329736406 + 0.5 * xs * xs + log(MIX_SIGMA[c]));
        mx = wsp[c] > mx ? wsp[c] : mx;
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) { wsp[c] = exp(wsp[c] - mx); nc += wsp[c]; }
      double probsum = 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c) { wsp[c] /= nc; probsum += wsp[c]; }
      // rmulti_mt (distributions/rmulti.cpp:41-78)
      const double tmp = d_runif(rng, 0.0, probsum);
      double psum = 0.0, sig = MIX_SIGMA[8];
      bool found = false;
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        psum += wsp[c];
        if (!found && tmp <= psum) { found = true; sig = MIX_SIGMA[c]; }
      }
      const double wgt = 1.0 / (sig * sig);
      info += wgt;
      sum += latent * wgt;
    }
  }
  P.z[(size_t)chain * P.n + i] = sum;
  P.w[(size_t)chain * P.n + i] = info;
}

// ---- Polya-Gamma augmentation (BASELINE config 5 as worded) ---------------------------
// BOOM has no Polya-Gamma sampler: this is the published algorithm --
// Polson, Scott and Windle (2013), JASA 108, sec. 4 / supplement Algorithm 1: PG(1, z) =
// J*(1, z / 2) / 4 by Devroye's alternating-series method with a truncated
// inverse-Gaussian proposal below t = 0.64 and an exponential one above -- with the
// oracle's bo_logit (imputer = 1) as its CPU twin, draw for draw.  omega_i ~ PG(n_i,
// x_i'beta) is the observation's information, kappa_i = y_i - n_i / 2 its
// information-weighted response: the same (sum, information) pair the auxiliary-mixture
// imputer produces, so everything after the imputation is the logit path unchanged.
// Observation i of sweep s reads the chain's stream 10 from position (s n + i) * PG_STRIDE.
namespace {
constexpr double PG_T = 0.64, PG_PI = 3.14159265358979323846;
__device__ __forceinline__ double pg_pnorm(double x) { return 0.5 * erfc(-x / 1.4142135623730951); }
__device__ __forceinline__ double pg_a(int n, double x) {
  const double K = (n + 0.5) * PG_PI;
  if (x > PG_T) return K * exp(-0.5 * K * K * x);
  const double expnt = -1.5 * (log(0.5 * PG_PI) + log(x)) + log(K) - 2.0 * (n + 0.5) * (n + 0.5) / x;
  return exp(expnt);
}
__device__ __forceinline__ double pg_rtigauss(SeqRng &r, double z, int *bad) {
  const double t = PG_T;
  double X = t + 1.0;
  if (z < 1.0 / t) {   // mu = 1 / z > t (z = 0: the Levy limit, alpha = 1)
    double alpha = 0.0;
    int it = 0;
    while (r() > alpha) {
      double E1 = d_exp_rand(r), E2 = d_exp_rand(r);
      while (E1 * E1 > 2 * E2 / t) { E1 = d_exp_rand(r); E2 = d_exp_rand(r); }
      X = 1 + E1 * t;
      X = t / (X * X);
      alpha = exp(-0.5 * z * z * X);
      if (++it > 10000) { *bad = 1; return t; }
    }
  } else {
    const double mu = 1.0 / z;
    int it = 0;
    while (X > t) {
      double Y = d_norm_rand(r);
      Y *= Y;
      const double half_mu = 0.5 * mu, mu_Y = mu * Y;
      X = mu + half_mu * mu_Y - half_mu * sqrt(4 * mu_Y + mu_Y * mu_Y);
      if (r() > mu / (mu + X)) X = mu * mu / X;
      if (++it > 10000) { *bad = 1; return t; }
    }
  }
  return X;
}
__device__ __forceinline__ double pg_draw1(SeqRng &r, double z, int *bad) {
  z = fabs(z) * 0.5;
  const double t = PG_T;
  const double fz = 0.125 * PG_PI * PG_PI + 0.5 * z * z;
  for (int tries = 0; tries < 10000; ++tries) {
    double X;
    {
      const double b = sqrt(1.0 / t) * (t * z - 1), a = -1.0 * sqrt(1.0 / t) * (t * z + 1);
      const double x0 = log(fz) + fz * t;
      const double xb = x0 - z + log(pg_pnorm(b)), xa = x0 + z + log(pg_pnorm(a));
      const double qdivp = 4 / PG_PI * (exp(xb) + exp(xa));
      if (r() < 1.0 / (1.0 + qdivp)) X = t + d_exp_rand(r) / fz;
      else X = pg_rtigauss(r, z, bad);
    }
    if (*bad) return 0.25 * X;
    double S = pg_a(0, X);
    const double Y = r() * S;
    int n = 0;
    bool go = true;
    while (go) {
      ++n;
      if (n & 1) {
        S -= pg_a(n, X);
        if (Y <= S) return 0.25 * X;
      } else {
        S += pg_a(n, X);
        if (Y > S) go = false;
      }
      if (n > 1000) { *bad = 1; return 0.25 * X; }
    }
  }
  *bad = 1;
  return 0.0;
}
}  // namespace

__global__ __launch_bounds__(256) void logit_pg_impute_kernel(ProbitParams P) {
  const int chain = (int)blockIdx.y, i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  __shared__ int s_status;   // (one decision per workgroup: see probit_impute_kernel)
  if (threadIdx.x == 0) s_status = __atomic_load_n(P.status + chain, __ATOMIC_RELAXED);
  __syncthreads();
  if (s_status != CHAIN_OK) return;
  __shared__ int s_idx[PROBIT_KMAX];
  __shared__ double s_beta[PROBIT_KMAX];
  const int k = included_coefficients(P, chain, s_idx, s_beta);
  if (k > PROBIT_KMAX) {
    if (threadIdx.x == 0 && blockIdx.x == 0) P.status[chain] = CHAIN_MODEL_TOO_LARGE;
    return;
  }
  if (i >= P.n) return;
  double eta = 0.0;
  for (int m = 0; m < k; ++m) eta += P.X[(size_t)s_idx[m] * P.n + i] * s_beta[m];
  const long nt = lround(P.ntrials[i]), ys = lround(P.y[i]);
  SeqRng rng = SeqRng::slot(PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 10u},
                            P.sweep * (uint64_t)P.n + (uint64_t)i, PG_STRIDE, slot_serve(P, PG_STRIDE));
  int bad = 0;
  double omega = 0.0;
  if (nt > P.clt_threshold) {
    // the normal with PG(n, z)'s mean n tanh(z/2) / (2 z) and variance
    // n (sinh z - z) / (4 z^3 cosh^2(z/2))
    const double az = fabs(eta);
    double mean, var;
    if (az < 1e-4) {
      mean = nt * (0.25 - az * az / 48.0);
      var = nt * (1.0 / 24.0 - az * az / 120.0);
    } else {
      const double th = tanh(0.5 * az), ch = cosh(0.5 * az);
      mean = nt * th / (2 * az);
      var = nt * (sinh(az) - az) / (4 * az * az * az * ch * ch);
    }
    omega = d_rnorm(rng, mean, sqrt(var));
    if (!(omega > 0)) omega = mean;
  } else {
    for (long t = 0; t < nt; ++t) omega += pg_draw1(rng, eta, &bad);
  }
  if (bad || rng.overran()) P.status[chain] = CHAIN_RNG_BRANCH;
  P.z[(size_t)chain * P.n + i] = (double)ys - 0.5 * (double)nt;
  P.w[(size_t)chain * P.n + i] = omega;
}

// ---- PoissonRegressionSpikeSlabSampler's imputation (the Poisson member)
// PoissonRegressionDataImputer::impute_latent_data_point (PoissonRegressionAuxMixSampler.cpp:
// 59-81) / PoissonDataImputer::impute (PoissonDataImputer.cpp:36-96): the time of the last
// of the y events in the exposure interval (exposure x Beta(y, 1), Cheng's algorithm BC,
// Bmath/rbeta.cpp:59-128), the first event past the interval (an exponential, or its
// extreme-value form when exp(eta) would overflow), and for both negative log times the
// component of the normal mixture that approximates NegLogGamma(count) (unmix:
// NormalMixtureApproximation.cpp:280-290; the mixtures are the reference table's, handed
// in by the caller).  z[chain][i] = sum of weight x (time - component mean), w[chain][i] =
// sum of weights: the (sum, information) pair of the logit path.  Observation i of sweep s
// reads the chain's stream 11 from position (s n + i) * POISSON_STRIDE.
namespace {
// Rmath::rbeta_mt(rng, aa, 1), aa >= 1: a = min = 1, b = aa, algorithm BC
__device__ __forceinline__ double d_rbeta_a_1(SeqRng &rng, double aa) {
  const double expmax = 1024 * 0.693147180559945309417232121458;
  const double a = (aa < 1.0) ? aa : 1.0, b = (aa < 1.0) ? 1.0 : aa;
  const double alpha = a + b;
  const double beta = 1.0 / a, delta = 1.0 + b - a;
  const double k1 = delta * (0.0138889 + 0.0416667 * a) / (b * beta - 0.777778);
  const double k2 = 0.25 + (0.5 + 0.25 / delta) * a;
  double u1, u2, v, w, y, z;
  for (;;) {
    u1 = rng();
    u2 = rng();
    if (u1 < 0.5) {
      y = u1 * u2;
      z = u1 * y;
      if (0.25 * u2 + z - y >= k1) continue;
    } else {
      z = u1 * u1 * u2;
      if (z <= 0.25) {
        v = beta * log(u1 / (1.0 - u1));
        w = (v <= expmax) ? b * exp(v) : 1.7976931348623157e308;
        break;
      }
      if (z >= k2) continue;
    }
    v = beta * log(u1 / (1.0 - u1));
    w = (v <= expmax) ? b * exp(v) : 1.7976931348623157e308;
    if (alpha * (log(alpha / (a + w)) + v) - 1.3862944 >= log(z)) break;
  }
  const double ans = (aa == a) ? a / (a + w) : w / (a + w);
  if (ans != ans) {
    const double zero = 2.220446049250313e-16, one = 1.0 - zero;
    if (aa == a) return isfinite(a) ? zero : one;
    return isfinite(w) ? zero : one;
  }
  return ans;
}
// unmix_poisson_augmented_data: mixture `mix` (-1: the Gaussian limit of `nevents` events)
__device__ __forceinline__ void poisson_unmix(const ProbitParams &P, SeqRng &rng, double u, int mix,
                                              double nevents, double *mu, double *sigsq, int *bad) {
  if (mix < 0) {
    *mu = -log(nevents);
    *sigsq = 1.0 / nevents;
    return;
  }
  const int c0 = P.mix_off[mix], nc = P.mix_off[mix + 1] - c0;
  double wsp[POISSON_MAX_COMP];
  double mx = -__builtin_huge_val(), tot = 0.0;
  for (int c = 0; c < nc; ++c) {
    const double sg = P.mix_sigma[c0 + c];
    const double xs = (u - P.mix_mu[c0 + c]) / sg;
    wsp[c] = P.mix_logw[c0 + c] + -(0.918938533204672741780329736406 + 0.5 * xs * xs + log(sg));
    mx = wsp[c] > mx ? wsp[c] : mx;
  }
  for (int c = 0; c < nc; ++c) { wsp[c] = exp(wsp[c] - mx); tot += wsp[c]; }
  double probsum = 0.0;
  for (int c = 0; c < nc; ++c) { wsp[c] /= tot; probsum += wsp[c]; }
  // rmulti_mt (distributions/rmulti.cpp:41-78)
  const double tmp = d_runif(rng, 0.0, probsum);
  double psum = 0.0;
  int ind = -1;
  for (int c = 0; c < nc; ++c) {
    psum += wsp[c];
    if (ind < 0 && tmp <= psum) ind = c;
  }
  if (ind < 0) { *bad = 1; ind = nc - 1; }
  const double sg = P.mix_sigma[c0 + ind];
  *mu = P.mix_mu[c0 + ind];
  *sigsq = sg * sg;
}
}  // namespace

__global__ __launch_bounds__(256) void poisson_impute_kernel(ProbitParams P) {
  const int chain = (int)blockIdx.y, i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  __shared__ int s_status;   // (one decision per workgroup: see probit_impute_kernel)
  if (threadIdx.x == 0) s_status = __atomic_load_n(P.status + chain, __ATOMIC_RELAXED);
  __syncthreads();
  if (s_status != CHAIN_OK) return;
  __shared__ int s_idx[PROBIT_KMAX];
  __shared__ double s_beta[PROBIT_KMAX];
  const int k = included_coefficients(P, chain, s_idx, s_beta);
  if (k > PROBIT_KMAX) {
    if (threadIdx.x == 0 && blockIdx.x == 0) P.status[chain] = CHAIN_MODEL_TOO_LARGE;
    return;
  }
  if (i >= P.n) return;
  double eta = 0.0;
  for (int m = 0; m < k; ++m) eta += P.X[(size_t)s_idx[m] * P.n + i] * s_beta[m];
  const long long y = llround(P.y[i]);
  const double exposure = P.ntrials[i];
  SeqRng rng = SeqRng::slot(PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 11u},
                            P.sweep * (uint64_t)P.n + (uint64_t)i, POISSON_STRIDE, slot_serve(P, POISSON_STRIDE));
  int bad = 0;
  const double t_final = y > 0 ? exposure * d_rbeta_a_1(rng, (double)y) : 0.0;
  const double delta = exposure - t_final;
  double z_ext;
  if (fabs(eta) < 600) {
    z_ext = -log(delta + (1.0 / exp(eta)) * d_exp_rand(rng));
  } else if (delta > 0) {
    const double err = -log((1.0 / 1.0) * d_exp_rand(rng)) * 1.0 + 0.0;   // rexv_mt(rng, 0, 1)
    const double xx = log(delta), yy = -err - eta;
    const double hi2 = xx < yy ? yy : xx, lo2 = xx < yy ? xx : yy;
    z_ext = -(hi2 + log1p(exp(lo2 - hi2)));                                // -lse2
  } else {
    z_ext = eta + (-log((1.0 / 1.0) * d_exp_rand(rng)) * 1.0 + 0.0);
  }
  double mu_e, sig_e, mu_i = 0.0, sig_i = 1.0, z_int = 0.0;
  poisson_unmix(P, rng, z_ext - eta, P.mix_one, 1.0, &mu_e, &sig_e, &bad);
  if (y > 0) {
    z_int = -log(t_final);
    poisson_unmix(P, rng, z_int - eta, P.obs_mix[i], (double)y, &mu_i, &sig_i, &bad);
  }
  // the internal point first, then the external one (the order the reference adds them in)
  double sum = 0.0, info = 0.0;
  if (y > 0) {
    const double w = 1.0 / sig_i;
    sum += w * (z_int - mu_i);
    info += w;
  }
  {
    const double w = 1.0 / sig_e;
    sum += w * (z_ext - mu_e);
    info += w;
  }
  if (bad || rng.overran()) P.status[chain] = CHAIN_RNG_BRANCH;
  P.z[(size_t)chain * P.n + i] = sum;
  P.w[(size_t)chain * P.n + i] = info;
}

hipError_t launch_rows_times_columns(hipStream_t stream, const double *U, int R, const double *B, int64_t n,
                                     int p, const double *diag_base, double *out, double *planes);

// impute + X'z for every chain
hipError_t launch_probit_impute(hipStream_t stream, const ProbitParams &P, double *planes) {
  hipError_t err;
  {
    KtScope kt(stream, KT_PROBIT_IMPUTE);
    hipLaunchKernelGGL(probit_impute_kernel, dim3((P.n + 255) / 256, P.chains), dim3(256), 0, stream, P);
    err = hipGetLastError();
  }
  if (err != hipSuccess) return err;
  return launch_rows_times_columns(stream, P.z, P.chains, P.X, (int64_t)P.n, P.p, nullptr, P.xtz, planes);
}


// impute, X'Wz and the diagonal of V = slab precision + X'WX for every chain (the rest
// of V is built a vector at a time, as the sweep asks for it: xtwx_cols_kernel.hip)
hipError_t launch_logit_impute(hipStream_t stream, const ProbitParams &P, const double *Xsq,
                               const double *slab_precision, double *v_diag, double *planes,
                               int polya_gamma) {
  hipError_t err;
  {
    KtScope kt(stream, polya_gamma == 2 ? KT_POISSON_IMPUTE : KT_LOGIT_IMPUTE);
    if (polya_gamma == 2)   // (the Poisson member of the family: same outputs, its own imputation)
      hipLaunchKernelGGL(poisson_impute_kernel, dim3((P.n + 255) / 256, P.chains), dim3(256), 0, stream, P);
    else if (polya_gamma)
      hipLaunchKernelGGL(logit_pg_impute_kernel, dim3((P.n + 255) / 256, P.chains), dim3(256), 0, stream, P);
    else
      hipLaunchKernelGGL(logit_impute_kernel, dim3((P.n + 255) / 256, P.chains), dim3(256), 0, stream, P);
    err = hipGetLastError();
  }
  if (err != hipSuccess) return err;
  err = launch_rows_times_columns(stream, P.z, P.chains, P.X, (int64_t)P.n, P.p, nullptr, P.xtz, planes);
  if (err != hipSuccess) return err;
  return launch_rows_times_columns(stream, P.w, P.chains, Xsq, (int64_t)P.n, P.p, slab_precision, v_diag, planes);
}

}  // namespace boom_amd
