// bsts "local level + regression": the state half of
// StateSpacePosteriorSampler::draw() for many chains, one chain per wavefront.
//
//   ZeroMeanGaussianConjSampler::draw      (ZeroMeanGaussianConjSampler.cpp:57-60)
//   Base::impute_state                     (StateSpaceModelBase.cpp:278-291)
//     clear_client_data                    (:248-254)
//     ScalarBase::simulate_forward         (:771-790)   data filter + simulation
//       ScalarMarginalDistribution::update (ScalarKalmanFilter.cpp:41-83)
//     Base::propagate_disturbances         (:858-891)
//       fast_disturbance_smooth            (ScalarKalmanFilter.cpp:168-196)
//     observe_state / observe_data_given_state
//       (LocalLevelStateModel.cpp:52-58, StateSpaceRegressionModel.cpp:188-200)
//
// State dimension 1 (Z = 1, T = 1, RQR = sigma^2_level), so every per-time-step
// quantity is a scalar.  The recursions are sequential in t: they run as
// wave-uniform scalar code over chunks of 64 time steps whose inputs sit one per
// lane (picked with v_readlane, results put back with a lane select), while
// everything that is independent across t (adjusted observations y - x'beta,
// v/F, residuals, X'r) is lane-parallel with coalesced reads of the shared
// column-major design matrix.  The stream of normals is consumed in the
// reference's order: state error then observation for every t.
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "kalman_params.h"

namespace boom_amd {

namespace {

constexpr int WAVE = 64;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x, double fill) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned long long f = __builtin_bit_cast(unsigned long long, fill);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double bcast_u(double x, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int bcast_u(int x, int src) {
  return __builtin_amdgcn_readlane(x, src);
}
__device__ __forceinline__ double wave_sum(double x) {
  x += dpp_f64<0x118, 0xf>(x, 0.0);
  x += dpp_f64<0x114, 0xf>(x, 0.0);
  x += dpp_f64<0x112, 0xf>(x, 0.0);
  x += dpp_f64<0x111, 0xf>(x, 0.0);
  x += dpp_f64<0x142, 0xa>(x, 0.0);
  x += dpp_f64<0x143, 0xc>(x, 0.0);
  return bcast_u(x, 63);
}

// Sequential reader of standard normals from a Philox stream: every lane
// evaluates the Kinderman-Ramage transform at its own offset of a 64-uniform
// window; next() walks the window the way a sequential reader would.
struct NormalStream {
  PhiloxKey key;
  uint64_t pos;  // stream position of the next draw
  double v;      // this lane's speculative draw (window offset = lane)
  int used;      // uniforms it consumed
  int cur;       // window offset of the next draw (>= 64: window exhausted)
  int lane;
  __device__ __forceinline__ void init(const PhiloxKey &k, uint64_t p, int l) {
    key = k; pos = p; lane = l; cur = WAVE; v = 0.0; used = 0;
  }
  __device__ __forceinline__ double next() {
    if (cur >= WAVE) {
      SeqRng r{key, pos + (uint64_t)lane};
      v = d_norm_rand(r);
      used = (int)(r.pos - (pos + (uint64_t)lane));
      cur = 0;
    }
    const double z = bcast_u(v, cur);
    const int adv = bcast_u(used, cur);
    cur += adv;
    pos += (uint64_t)adv;
    return z;
  }
  // rnorm_mt(mu, sigma): no draw when sigma == 0 (Bmath/rnorm.cpp:63-64)
  __device__ __forceinline__ double rnorm(double mu, double sigma) {
    if (sigma == 0.0) return mu;
    return mu + sigma * next();
  }
};

}  // namespace

// grid = chains, block = 64
__global__ __launch_bounds__(64) void kalman_simsmooth_kernel(SsParams P,
                                                              int draw_level) {
  const int chain = blockIdx.x, lane = threadIdx.x;
  if (chain >= P.chains) return;
  if (P.status[chain] != CHAIN_OK) return;
  const int T = P.T, p = P.p;
  int status = CHAIN_OK;

  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  double level_sigsq = P.level_sigsq[chain];

  // ---- ZeroMeanGaussianConjSampler::draw: sigma^2_level | state
  if (draw_level) {
    SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 1u}, P.pos_level[chain]};
    int bad = 0;
    const double DF = P.level_n[chain] + P.level_prior_df;
    const double SS = P.level_sumsq[chain] + P.level_prior_ss;
    level_sigsq = d_draw_variance(rng, DF, SS, P.level_sigma_max, &bad);
    if (bad) status = CHAIN_RNG_BRANCH;
    if (lane == 0) {
      P.pos_level[chain] = rng.pos;
      P.level_sigsq[chain] = level_sigsq;
    }
  }
  if (status != CHAIN_OK) {
    if (lane == 0) P.status[chain] = status;
    return;
  }

  // ---- impute_state ------------------------------------------------------
  const double sigsq_obs = P.sigsq[chain];
  const double *beta = P.beta + (size_t)chain * p;
  const uint8_t *gamma = P.gamma + (size_t)chain * p;
  double *sv = P.scratch + (size_t)chain * P.scratch_stride;  // v  (data filter)
  double *sF = sv + T;                                        // F
  double *sK = sF + T;                                        // K
  double *svs = sK + T;                                       // v  (simulation)
  double *sst = svs + T;                                      // state draw
  double *sr = sst + T;                                       // r  (data)
  double *srs = sr + T;                                       // r  (simulation)

  // adjusted observations y*_t = y_t - x_t'beta, lane-parallel over t
  // (StateSpaceRegressionModel.cpp:65-77, 179-181; GlmCoefs::predict is a
  // dense dot with Beta(), zeros outside gamma)
  {
    int nvars = 0;
    for (int base = 0; base < p; base += WAVE) {
      const int j = base + lane;
      nvars += __popcll(__ballot(j < p && gamma[j] != 0));
    }
    for (int t0 = 0; t0 < T; t0 += WAVE) {
      const int t = t0 + lane;
      double pred = 0.0;
      if (nvars > 0 && t < T) {
        for (int j = 0; j < p; ++j) {
          const double bj = beta[j];
          if (bj != 0.0) pred += P.X[(size_t)j * T + t] * bj;
        }
      }
      if (t < T) sv[t] = (P.y[t] - pred) / 1;
    }
  }
  __syncthreads();

  // forward pass: variance recursion, data filter, simulation + its filter
  const double q = level_sigsq;
  const double level_sigma = sqrt(level_sigsq);
  const double H = sigsq_obs;  // one observation per time point (n_t = 1)
  const double sqrtH = sqrt(H);
  NormalStream ns;
  ns.init(PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, P.pos_state[chain], lane);
  double Pv = P.P0, a = P.a0, as = P.a0, alpha = 0.0;
  for (int t0 = 0; t0 < T && status == CHAIN_OK; t0 += WAVE) {
    const int t = t0 + lane;
    const double yreg = (t < T) ? sv[t] : 0.0;
    const unsigned long long obsmask = __ballot(t < T && P.observed[t] != 0);
    double vreg = 0, Freg = 0, Kreg = 0, vsreg = 0, streg = 0;
    const int nthis = (T - t0 < WAVE) ? (T - t0) : WAVE;
    for (int i = 0; i < nthis; ++i) {
      const bool miss = !((obsmask >> i) & 1ull);
      const double y = bcast_u(yreg, i);
      // ScalarMarginalDistribution::update
      const double PZ = Pv;
      const double F = PZ + H;
      if (!(F > 0.0)) { status = CHAIN_FORECAST_VARIANCE; break; }
      const double K = miss ? 0.0 : PZ / F;
      const double v = miss ? 0.0 : y - a;
      if (!miss) a = a + K * v;
      // simulate_initial_state / simulate_next_state, then
      // simulate_adjusted_observation
      if (t0 + i == 0) alpha = ns.rnorm(P.a0, sqrt(P.P0));
      else alpha = alpha + ns.rnorm(0.0, level_sigma);
      const double ysim = ns.rnorm(alpha, sqrtH);
      const double vs = miss ? 0.0 : ysim - as;
      if (!miss) as = as + K * vs;
      if (!miss) Pv = Pv + (-1.0) * PZ * K;
      Pv = Pv + q;
      if (lane == i) { vreg = v; Freg = F; Kreg = K; vsreg = vs; streg = alpha; }
    }
    if (t < T) { sv[t] = vreg; sF[t] = Freg; sK[t] = Kreg; svs[t] = vsreg; sst[t] = streg; }
  }
  if (status != CHAIN_OK) {
    if (lane == 0) P.status[chain] = status;
    return;
  }
  __syncthreads();

  // backward pass: fast_disturbance_smooth for both filters
  double r = 0.0, rs = 0.0;
  for (int t0 = ((T - 1) / WAVE) * WAVE; t0 >= 0; t0 -= WAVE) {
    const int t = t0 + lane;
    const bool in = t < T;
    const double F = in ? sF[t] : 1.0, K = in ? sK[t] : 0.0;
    const double u = in ? sv[t] / F : 0.0, us = in ? svs[t] / F : 0.0;
    double rreg = 0.0, rsreg = 0.0;
    const int nthis = (T - t0 < WAVE) ? (T - t0) : WAVE;
    for (int i = nthis - 1; i >= 0; --i) {
      const double Ki = bcast_u(K, i);
      const double c = bcast_u(u, i) - Ki * r;
      const double cs = bcast_u(us, i) - Ki * rs;
      if (lane == i) { rreg = r; rsreg = rs; }
      r = r + c;
      rs = rs + cs;
    }
    if (in) { sr[t] = rreg; srs[t] = rsreg; }
  }
  __syncthreads();

  // forward mean correction + level sufficient statistics (sequential sums in
  // the reference's order)
  double mean_sim = P.a0 + P.P0 * rs;
  double mean_obs = P.a0 + P.P0 * r;
  double lev_n = 0.0, lev_ss = 0.0, prev_state = 0.0;
  for (int t0 = 0; t0 < T; t0 += WAVE) {
    const int t = t0 + lane;
    const bool in = t < T;
    const double rprev = (in && t > 0) ? sr[t - 1] : 0.0;
    const double rsprev = (in && t > 0) ? srs[t - 1] : 0.0;
    const double st0 = in ? sst[t] : 0.0;
    double streg = 0.0;
    const int nthis = (T - t0 < WAVE) ? (T - t0) : WAVE;
    for (int i = 0; i < nthis; ++i) {
      if (t0 + i > 0) {
        mean_sim = mean_sim + q * bcast_u(rsprev, i);
        mean_obs = mean_obs + q * bcast_u(rprev, i);
      }
      const double s = bcast_u(st0, i) + (mean_obs - mean_sim);
      if (t0 + i > 0) {
        const double diff = s - prev_state;
        lev_n += 1.0;
        lev_ss += diff * diff;
      }
      prev_state = s;
      if (lane == i) streg = s;
    }
    if (in) sst[t] = streg;
  }
  __syncthreads();

  // regression sufficient statistics given the state
  // (observe_data_given_state + NeRegSuf::add_mixture_data): residual
  // e_t = y_t - alpha_t on observed t; xty = X'e, yty = e'e, n = #observed
  double part_q = 0.0, part_n = 0.0;
  for (int t0 = 0; t0 < T; t0 += WAVE) {
    const int t = t0 + lane;
    double e = 0.0;
    if (t < T && P.observed[t]) {
      e = P.y[t] - sst[t];
      part_q += e * e;
      part_n += 1.0;
    }
    if (t < T) sv[t] = e;  // residual, zero where unobserved
  }
  const double yty = wave_sum(part_q);
  const double nobs = wave_sum(part_n);
  __syncthreads();
  double *xty = P.xty + (size_t)chain * p;
  for (int j = 0; j < p; ++j) {
    const double *col = P.X + (size_t)j * T;
    double acc = 0.0;
    for (int t = lane; t < T; t += WAVE) acc += col[t] * sv[t];
    const double tot = wave_sum(acc);
    if (lane == 0) xty[j] = tot;
  }
  if (lane == 0) {
    P.yty[chain] = yty;
    P.nobs[chain] = nobs;
    P.level_n[chain] = lev_n;
    P.level_sumsq[chain] = lev_ss;
    P.pos_state[chain] = ns.pos;
    P.status[chain] = status;
  }
}

hipError_t launch_kalman_simsmooth(hipStream_t stream, const SsParams &P,
                                   int draw_level) {
  hipLaunchKernelGGL(kalman_simsmooth_kernel, dim3(P.chains), dim3(WAVE), 0,
                     stream, P, draw_level);
  return hipGetLastError();
}

}  // namespace boom_amd
