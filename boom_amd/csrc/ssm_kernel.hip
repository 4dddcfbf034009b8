// bsts structural time series for many chains: the state half of
// StateSpacePosteriorSampler::draw() when the state is a trend block
// (LocalLevelStateModel, or LocalLinearTrendStateModel with one
// ZeroMeanMvnIndependenceSampler per variance) plus an optional
// SeasonalStateModel(nseasons, season_duration = 1).  SURVEY 8f row f2.
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
//                    SeasonalStateModel.cpp:74-86), observe_data_given_state
//
// State vector [trend (1 or 2) | seasonal (nseasons - 1)], dimension m <= 16.
//   Z    ones at the first element of each block
//   T    trend [1] or [[1, 1], [0, 1]]; seasonal: first row -1, ones below the diagonal
//   RQR  diagonal: level, slope and the seasonal block's first element
// One chain per workgroup of two wavefronts: both share the adjusted
// observations and the sweep's normals (stream_normals.h), then wave 0 runs the
// three passes over time.  Lane j < m holds component j of every state-sized
// vector and column j of the state variance P (16 registers).  Unlike the
// local-level kernel (kalman_kernel.hip) the passes are SERIAL in time: the
// per-step maps are m x m here and their compositions no longer fit a wave
// scan.  As there, the data filter and the simulation filter share the gains, so
// ONE filter runs on w = y* - y+, and one smoother on the difference.
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
  __device__ __forceinline__ bool seasonal(int i) const { return ns > 0 && i >= s0 && i < s0 + ns; }
};

// y = T x for a vector held one component per lane
__device__ __forceinline__ double vecT(const Shape &S, double x, int lane) {
  double y = x;
  if (S.trend == 2) {
    const double x1 = rl(x, 1);
    if (lane == 0) y = x + x1;
  }
  if (S.ns > 0) {
    const double tot = row_total(S.seasonal(lane) ? x : 0.0);
    const double prev = sdpp<0x111, 0xf>(x, 0.0);   // lane - 1
    if (lane == S.s0) y = -tot;
    else if (S.seasonal(lane)) y = prev;
  }
  return y;
}
// y = T' x
__device__ __forceinline__ double vecTt(const Shape &S, double x, int lane) {
  double y = x;
  if (S.trend == 2) {
    const double x0 = rl(x, 0);
    if (lane == 1) y = x0 + x;
  }
  if (S.ns > 0) {
    const double first = rl(x, S.s0);
    const double next = sdpp<0x101, 0xf>(x, 0.0);   // row_shl:1 = lane + 1
    if (S.seasonal(lane)) y = -first + ((lane + 1 < S.s0 + S.ns) ? next : 0.0);
  }
  return y;
}
__device__ __forceinline__ double zdot(const Shape &S, double x) {
  double a = rl(x, 0);
  if (S.ns > 0) a += rl(x, S.s0);
  return a;
}

}  // namespace

// grid = chains, block = 128
__global__ __launch_bounds__(128) void ssm_simsmooth_kernel(SsParams P, int draw_variances) {
  __shared__ NormalsLds s_norm;
  __shared__ double s_tr[SSM_MAX][SSM_MAX + 1];   // transposition of P
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) return;
  if (P.only_ran && P.only_ran[chain] == 0) return;
  const SsmParams &Q = P.ssm;
  const int T = P.T, p = P.p, m = Q.m;
  Shape S;
  S.m = m; S.trend = Q.trend; S.s0 = Q.s0; S.ns = Q.nseasons > 0 ? Q.nseasons - 1 : 0;
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  int status = CHAIN_OK;

  // ---- the state models' variance draws, in model order: level [, slope], seasonal
  double sig2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) sig2[i] = Q.var_sigsq[(size_t)chain * 3 + i];
  if (draw_variances) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool active = (i == 0) || (i == 1 && S.trend == 2) || (i == 2 && S.ns > 0);
      if (!active) continue;
      const uint32_t sid = (i == 0) ? 1u : (i == 1 ? 6u : 7u);
      SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, sid}, Q.pos_var[(size_t)chain * 3 + i]};
      int bad = 0;
      const double DF = Q.var_n[(size_t)chain * 3 + i] + Q.prior_df[i];
      const double SSQ = Q.var_ss[(size_t)chain * 3 + i] + Q.prior_ss[i];
      double draw = d_draw_variance(rng, DF, SSQ, Q.sigma_max[i], &bad);
      if (bad) status = CHAIN_RNG_BRANCH;
      // ZeroMeanMvnIndependenceSampler sets siginv(i, i) = 1 / draw; the model's
      // Sigma is the inverse of that again
      if (S.trend == 2 && i < 2) draw = 1.0 / (1.0 / draw);
      sig2[i] = draw;
      if (lane == 0 && wave == 0) {
        Q.pos_var[(size_t)chain * 3 + i] = rng.pos;
        Q.var_sigsq[(size_t)chain * 3 + i] = draw;
      }
    }
  }
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }

  const double H = P.sigsq[chain], sqrtH = sqrt(H);
  const double sdv[3] = {sqrt(sig2[0]), sqrt(sig2[1]), sqrt(sig2[2])};
  const double *beta = P.beta + (size_t)chain * p;
  double *w0 = P.scratch + (size_t)chain * P.scratch_stride;   // y* -> (v - v+) / F
  double *sres = w0 + T;                                       // residuals (input of the X'e GEMM)
  double *wk = Q.work + (size_t)chain * Q.work_stride;
  double *gK = wk;                                 // K_t, m per step
  double *gst = gK + (size_t)m * T;                // alpha+_t, then the state draw
  double *gd = gst + (size_t)m * T;                // r_t (difference) at the three rows with state error
  double *szz = gd + (size_t)3 * T;                // the sweep's normals

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

  // ---- 2. the normals of simulate_forward, in stream order.  t = 0: the initial
  // state of every state model (rmvn_mt draws every component; the local level
  // model draws rnorm_mt(a0, sd0): nothing if sd0 == 0), then the observation;
  // t >= 1: the state errors (local level: one if sigma != 0; local linear trend:
  // two, always; seasonal: one if sigma != 0), then the observation.
  const int dH = (sqrtH != 0.0);
  const int d0 = (S.trend == 1) ? (Q.P0[0] != 0.0 ? 1 : 0) : 2;
  const int nfirst = d0 + S.ns + dH;
  const int dT = (S.trend == 1) ? (sdv[0] != 0.0 ? 1 : 0) : 2;
  const int dS = (S.ns > 0 && sdv[2] != 0.0) ? 1 : 0;
  const int nper = dT + dS + dH;
  const int N = nfirst + (T - 1) * nper;
  status = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, P.pos_state[chain], N,
                          szz, &P.pos_state[chain]);
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }
  __syncthreads();
  if (wave != 0) return;

  // ---- 3. forward: simulate alpha+, y+; variances P_t -> F_t, K_t; the filter on w = y* - y+
  const bool mylane = lane < m;
  const double rqr = !mylane ? 0.0
                     : (lane == 0 ? sig2[0] : ((S.trend == 2 && lane == 1) ? sig2[1] : ((S.ns > 0 && lane == S.s0) ? sig2[2] : 0.0)));
  const double sd_lane = !mylane ? 0.0
                         : (lane == 0 ? sdv[0] : ((S.trend == 2 && lane == 1) ? sdv[1] : ((S.ns > 0 && lane == S.s0) ? sdv[2] : 0.0)));
  double a0l = 0.0, P0l = 0.0;
#pragma unroll
  for (int i = 0; i < SSM_MAX; ++i) if (lane == i) { a0l = Q.a0[i]; P0l = Q.P0[i]; }
  double Pc[SSM_MAX];   // column `lane` of P
#pragma unroll
  for (int i = 0; i < SSM_MAX; ++i) Pc[i] = (mylane && i == lane) ? P0l : 0.0;
  double alpha = 0.0, delta = 0.0;   // alpha+_t, a_t - a+_t
  bool steady = false;               // P repeats bitwise: F, K stay
  double Fst = 0.0, Kst = 0.0;
  for (int tb = 0; tb < T; tb += WAVE) {
    // this block's inputs, one step per lane
    const int tt = tb + lane;
    const bool in_l = tt < T;
    const double ys_l = in_l ? w0[tt] : 0.0;
    const int ob_l = (in_l && P.observed[tt]) ? 1 : 0;
    const int nb_l = (tt == 0) ? 0 : nfirst + (tt - 1) * nper;
    // normals of the step: trend (up to 2), seasonal, observation
    double z0_l = 0.0, z1_l = 0.0, zs_l = 0.0, zh_l = 0.0;
    if (in_l && tt > 0) {
      int o = nb_l;
      if (dT >= 1) z0_l = szz[o++];
      if (dT == 2) z1_l = szz[o++];
      if (dS) zs_l = szz[o++];
      if (dH) zh_l = szz[o];
    } else if (in_l) {
      if (dH) zh_l = szz[d0 + S.ns];
    }
    const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
    for (int s = 0; s < nstep; ++s) {
      const int t = tb + s;
      const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
      const double ystar = rl(ys_l, s), zh = rl(zh_l, s);
      // simulate_initial_state / simulate_next_state
      if (t == 0) {
        double z = 0.0;
        if (mylane) {
          if (lane < S.trend) z = (lane < d0) ? szz[lane] : 0.0;
          else z = szz[d0 + (lane - S.s0)];
        }
        alpha = mylane ? sqrt(P0l) * z + a0l : 0.0;
      } else {
        const double z0 = rl(z0_l, s), z1 = rl(z1_l, s), zs = rl(zs_l, s);
        const double zl = (lane == 0) ? z0 : ((S.trend == 2 && lane == 1) ? z1 : zs);
        alpha = vecT(S, alpha, lane) + sd_lane * zl;
      }
      const double yplus = zdot(S, alpha) + sqrtH * zh;   // simulate_adjusted_observation
      const double w = ystar - yplus;
      // ---- ScalarMarginalDistribution::update, the part that does not look at the data
      double F, K;
      if (steady && obs) {
        F = Fst; K = Kst;
      } else {
        steady = false;
        // PZ_i = P(i, 0) + P(i, s0) = P(0, i) + P(s0, i): P is kept exactly symmetric
        double PZ = Pc[0];
        if (S.ns > 0) {
#pragma unroll
          for (int i = 1; i < SSM_MAX; ++i) if (i == S.s0) PZ += Pc[i];
        }
        if (!mylane) PZ = 0.0;
        F = zdot(S, PZ) + H;
        const double TPZ = vecT(S, PZ, lane);
        K = obs ? TPZ / F : 0.0;
        // sandwich_inplace: T times every column ...
        double old[SSM_MAX];
#pragma unroll
        for (int i = 0; i < SSM_MAX; ++i) old[i] = Pc[i];
        if (S.trend == 2) Pc[0] = Pc[0] + Pc[1];
        if (S.ns > 0) {
          double first = 0.0;
#pragma unroll
          for (int i = 1; i < SSM_MAX; ++i) if (S.seasonal(i)) first -= Pc[i];
#pragma unroll
          for (int i = SSM_MAX - 1; i >= 2; --i) if (S.seasonal(i) && i > S.s0) Pc[i] = Pc[i - 1];
#pragma unroll
          for (int i = 1; i < SSM_MAX; ++i) if (i == S.s0) Pc[i] = first;
        }
        // ... then T times every row (row i lives in register i across the lanes)
#pragma unroll
        for (int i = 0; i < SSM_MAX; ++i) if (i < m) Pc[i] = vecT(S, mylane ? Pc[i] : 0.0, lane);
        // - TPZ K' (observed steps), + RQR
#pragma unroll
        for (int i = 0; i < SSM_MAX; ++i) {
          if (i < m) {
            const double tpz_i = rl(TPZ, i);
            if (obs) Pc[i] += -1.0 * tpz_i * K;
            if (i == lane) Pc[i] += rqr;
          }
        }
        // fix_near_symmetry: P(i, j) = P(j, i) = (P(i, j) + P(j, i)) / 2
        if (mylane) {
#pragma unroll
          for (int i = 0; i < SSM_MAX; ++i) if (i < m) s_tr[i][lane] = Pc[i];
        }
        wave_lds_sync();
        if (mylane) {
#pragma unroll
          for (int i = 0; i < SSM_MAX; ++i) if (i < m && i != lane) Pc[i] = .5 * (Pc[i] + s_tr[lane][i]);
        }
        wave_lds_sync();
        bool same = true;
#pragma unroll
        for (int i = 0; i < SSM_MAX; ++i)
          same = same && (__builtin_bit_cast(unsigned long long, Pc[i]) == __builtin_bit_cast(unsigned long long, old[i]));
        if (obs && __all(same || !mylane)) { steady = true; Fst = F; Kst = K; }
      }
      if (!(F > 0.0)) { status = CHAIN_FORECAST_VARIANCE; break; }
      // ---- the filter on w: v - v+ = w - Z'(a - a+); a - a+ <- T (a - a+) + K (v - v+)
      const double e = obs ? w - zdot(S, delta) : 0.0;
      delta = vecT(S, delta, lane) + K * e;
      if (mylane) {
        gK[(size_t)t * m + lane] = K;
        gst[(size_t)t * m + lane] = alpha;
      }
      if (lane == 0) w0[t] = obs ? e / F : 0.0;
    }
    if (status != CHAIN_OK) break;
  }
  if (status != CHAIN_OK) {
    if (lane == 0) P.status[chain] = status;
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  // ---- 4. backward: fast_disturbance_smooth for d = r - r+:
  // r_{t-1} = T' r_t + Z ((v_t - v+_t) / F_t - K_t' r_t), r_{T-1} = 0
  // (time runs in blocks of PB steps whose loads are issued together: a step's
  // inputs do not depend on the recursion, its latency must not either)
  constexpr int PB = 16;
  double r = 0.0;
  const int slot_l = (lane == 0) ? 0 : ((S.trend == 2 && lane == 1) ? 1 : 2);
  const bool err_lane = mylane && rqr != 0.0;
  for (int tb = ((T - 1) / PB) * PB; tb >= 0; tb -= PB) {
    double Kb[PB], efb[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int t = tb + j;
      Kb[j] = (mylane && t < T) ? gK[(size_t)t * m + lane] : 0.0;
      efb[j] = (t < T) ? w0[t] : 0.0;
    }
#pragma unroll
    for (int j = PB - 1; j >= 0; --j) {
      const int t = tb + j;
      if (t < T) {
        // r_t at the rows that carry state error: what the correction pass needs
        if (err_lane) gd[(size_t)t * 3 + slot_l] = r;
        const double kr = row_total(Kb[j] * r);
        const double coef = efb[j] - kr;
        r = vecTt(S, r, lane);
        if (lane == 0 || (S.ns > 0 && lane == S.s0)) r += coef;
        if (!mylane) r = 0.0;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  // ---- 5. forward: the mean correction E(alpha | y) - E(alpha | y+), the state
  // draw, the state models' and the regression's sufficient statistics
  double mc = P0l * r;          // a0 + P0 r0 - (a0 + P0 r0+)
  double prev = 0.0;            // state_{t-1}
  double suf_ss[3] = {0.0, 0.0, 0.0};
  double mv_ybar = 0.0, mv_sumsq = 0.0, mv_n = 0.0;   // MvnSuf of the trend errors (lanes 0, 1)
  double yty = 0.0, nobs = 0.0;
  for (int tb = 0; tb < T; tb += PB) {
    double ab[PB], db[PB], yb[PB];
    bool ob[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int t = tb + j;
      const bool in = t < T;
      ab[j] = (mylane && in) ? gst[(size_t)t * m + lane] : 0.0;
      db[j] = (err_lane && in && t > 0) ? gd[(size_t)(t - 1) * 3 + slot_l] : 0.0;
      yb[j] = in ? P.y[t] : 0.0;
      ob[j] = in && P.observed[in ? t : 0] != 0;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int t = tb + j;
      if (t >= T) break;
      if (t > 0) mc = vecT(S, mc, lane) + rqr * db[j];
      const double st = mylane ? ab[j] + mc : 0.0;
      if (t > 0) {
        if (S.trend == 1) {
          const double diff = st - prev;                 // (lane 0)
          if (lane == 0) suf_ss[0] += diff * diff;
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
        if (S.ns > 0) {
          // delta = now[0] + sum(then) over the seasonal block
          const double tot = row_total(S.seasonal(lane) ? prev : 0.0);
          const double dl = st - (-1.0 * tot);
          if (lane == S.s0) suf_ss[2] += dl * dl;
        }
      }
      prev = st;
      if (mylane) gst[(size_t)t * m + lane] = st;
      const double resid = ob[j] ? yb[j] - zdot(S, st) : 0.0;
      if (lane == 0) {
        sres[t] = resid;
        if (ob[j]) { yty += resid * resid; nobs += 1.0; }
      }
    }
  }
  // publish the sufficient statistics
  if (S.trend == 2) {
    // center_sumsq(mu = 0)(i, i) = sumsq_ii + n ybar_i^2
    const double ssv = mv_sumsq + mv_ybar * mv_ybar * mv_n;
    if (lane < 2) {
      Q.var_n[(size_t)chain * 3 + lane] = mv_n;
      Q.var_ss[(size_t)chain * 3 + lane] = ssv;
    }
  } else if (lane == 0) {
    Q.var_n[(size_t)chain * 3 + 0] = (double)(T - 1);
    Q.var_ss[(size_t)chain * 3 + 0] = suf_ss[0];
  }
  if (S.ns > 0 && lane == S.s0) {
    Q.var_n[(size_t)chain * 3 + 2] = (double)(T - 1);
    Q.var_ss[(size_t)chain * 3 + 2] = suf_ss[2];
  }
  if (lane == 0) {
    P.yty[chain] = yty;
    P.nobs[chain] = nobs;
    P.status[chain] = status;
  }
}

hipError_t launch_atb_mfma(hipStream_t stream, const double *A, int64_t lda, int M,
                           const double *B, int64_t ldb, int N, int K, double *C, int ldc);

hipError_t launch_ssm_simsmooth(hipStream_t stream, const SsParams &P, int draw_variances) {
  hipLaunchKernelGGL(ssm_simsmooth_kernel, dim3(P.chain_count), dim3(2 * WAVE), 0, stream, P,
                     draw_variances);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  // xty[chain, j] = x_j' e_chain (the residual series are array 1 of every chain's scratch block)
  return launch_atb_mfma(stream, P.scratch + (size_t)P.chain_first * P.scratch_stride + P.T, P.scratch_stride,
                         P.chain_count, P.X, (int64_t)P.T, P.p, P.T,
                         P.xty + (size_t)P.chain_first * P.p, P.p);
}

}  // namespace boom_amd
