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

#include "ktimer.h"

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
template <int TREND, bool SEAS>
__device__ __forceinline__ double vecT(const Shape &S, double x, int lane, int c) {
  double y = x;
  if (TREND == 2) {
    const double x1 = rl(x, 1);
    if (lane == 0) y = x + x1;
  }
  if (SEAS) {
    const double tot = row_total(S.seasonal(lane) ? x : 0.0);
    if (lane == S.s0 + cursor_prev(c, S.ns)) y = -tot;
  }
  return y;
}
// y = T' x; c1: cursor of x's layout (the result is in the previous step's)
template <int TREND, bool SEAS>
__device__ __forceinline__ double vecTt(const Shape &S, double x, int lane, int c1) {
  double y = x;
  if (TREND == 2) {
    const double x0 = rl(x, 0);
    if (lane == 1) y = x0 + x;
  }
  if (SEAS) {
    const double first = rl(x, S.s0 + c1);
    if (S.seasonal(lane)) y = (lane == S.s0 + c1) ? -first : x - first;
  }
  return y;
}
// Z'x, c: cursor of x's layout
template <bool SEAS>
__device__ __forceinline__ double zdot(const Shape &S, double x, int c) {
  double a = rl(x, 0);
  if (SEAS) a += rl(x, S.s0 + c);
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

}  // namespace

// grid = chains, block = 128.  TREND: 1 local level, 2 local linear trend; SEAS: a
// seasonal block follows
template <int TREND, bool SEAS>
__global__ __launch_bounds__(128) void ssm_simsmooth_kernel(SsParams P, int draw_variances) {
  // (the passes' buffers take the place of the normals generator's lists, which
  // are done by then: 37 KB per workgroup, four workgroups per CU)
  struct PassLds {
    double blk[2][WAVE * SSM_MAX];   // a block of 64 steps of a state-sized series, per wave
    double P[SSM_MAX * SSM_MAX];     // the state variance (wave 1)
    double tv[SSM_MAX];
  };
  union SharedLds {
    NormalsLds norm;
    PassLds pass;
  };
  __shared__ SharedLds s_lds;
  __shared__ int s_flag;
  double (&s_blk)[2][WAVE * SSM_MAX] = s_lds.pass.blk;
  double (&s_P)[SSM_MAX * SSM_MAX] = s_lds.pass.P;
  double (&s_tv)[SSM_MAX] = s_lds.pass.tv;
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) return;
  if (P.only_ran && P.only_ran[chain] == 0) return;
  const SsmParams &Q = P.ssm;
  const int T = P.T, p = P.p, m = Q.m;
  Shape S;
  S.m = m; S.trend = TREND; S.s0 = TREND; S.ns = SEAS ? Q.nseasons - 1 : 0;
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  int status = CHAIN_OK;
  if (threadIdx.x == 0) s_flag = CHAIN_OK;
#ifdef BA_KSTAMPS
  long long kph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, klast = (long long)__builtin_readcyclecounter();
#define SSTAMP(i) do { const long long t_ = (long long)__builtin_readcyclecounter(); kph[i] += t_ - klast; klast = t_; } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif

  // ---- the state models' variance draws, in model order: level [, slope], seasonal
  double sig2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) sig2[i] = Q.var_sigsq[(size_t)chain * 3 + i];
  if (draw_variances) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool active = (i == 0) || (i == 1 && TREND == 2) || (i == 2 && SEAS);
      if (active) {
        const uint32_t sid = (i == 0) ? 1u : (i == 1 ? 6u : 7u);
        SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, sid}, Q.pos_var[(size_t)chain * 3 + i]};
        int bad = 0;
        const double DF = Q.var_n[(size_t)chain * 3 + i] + Q.prior_df[i];
        const double SSQ = Q.var_ss[(size_t)chain * 3 + i] + Q.prior_ss[i];
        double draw = d_draw_variance(rng, DF, SSQ, Q.sigma_max[i], &bad);
        if (bad) status = CHAIN_RNG_BRANCH;
        // ZeroMeanMvnIndependenceSampler sets siginv(i, i) = 1 / draw; the model's
        // Sigma is the inverse of that again
        if (TREND == 2 && i < 2) draw = 1.0 / (1.0 / draw);
        sig2[i] = draw;
        if (lane == 0 && wave == 0) {
          Q.pos_var[(size_t)chain * 3 + i] = rng.pos;
          Q.var_sigsq[(size_t)chain * 3 + i] = draw;
        }
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
  double *w0 = P.scratch + (size_t)chain * P.scratch_stride;   // y* -> w = y* - y+ -> (v - v+) / F
  double *sres = w0 + T;                                       // F_t, then residuals (input of the X'e GEMM)
  double *wk = Q.work + (size_t)chain * Q.work_stride;
  double *gK = wk;                                 // K_t, m per step (layout of step t + 1)
  double *gst = gK + (size_t)m * T;                // alpha+_t (layout of step t), then the state draw
  double *gd = gst + (size_t)m * T;                // r_t (difference) at the three rows with state error: 3 series of T
  double *szz = gd + (size_t)3 * T;                // the sweep's normals

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
  // two, always; seasonal: one if sigma != 0), then the observation.
  const int dH = (sqrtH != 0.0);
  const int d0 = (TREND == 1) ? (Q.P0[0] != 0.0 ? 1 : 0) : 2;
  const int nfirst = d0 + S.ns + dH;
  const int dT = (TREND == 1) ? (sdv[0] != 0.0 ? 1 : 0) : 2;
  const int dS = (SEAS && sdv[2] != 0.0) ? 1 : 0;
  const int nper = dT + dS + dH;
  const int N = nfirst + (T - 1) * nper;
  status = stream_normals(s_lds.norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, P.pos_state[chain], N,
                          szz, &P.pos_state[chain]);
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
  for (int i = 0; i < SSM_MAX; ++i) if (lane == i) { a0l = Q.a0[i]; P0l = Q.P0[i]; }
  double *blk = s_blk[wave];

  // ---- 3. forward, the two waves side by side (neither needs the other's results):
  //   wave 1: the variances P_t -> F_t, K_t (ScalarMarginalDistribution::update, the
  //           part that does not look at the data);
  //   wave 0: simulate alpha+_t, y+_t and w_t = y*_t - y+_t.
  // Time runs in blocks of 64 steps: a block's scalar inputs sit one step per lane
  // (read with v_readlane), its state-sized series in LDS, and what a block
  // produces goes out in one coalesced piece.
  if (wave == 1) {
    // P lives in LDS (s_P[row * 16 + column], both indices in the rotating
    // layout, kept exactly symmetric): the rows and columns a step touches move
    // with the cursor, which registers cannot follow.  Lane k < m owns column k;
    // the rank-one update runs over all 256 entries on all 64 lanes.
    for (int e = lane; e < SSM_MAX * SSM_MAX; e += WAVE) s_P[e] = 0.0;
    if (lane < SSM_MAX) s_tv[lane] = 0.0;
    __builtin_amdgcn_wave_barrier();
    if (mylane) s_P[lane * (SSM_MAX + 1)] = P0l;
    __builtin_amdgcn_wave_barrier();
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const int ob_l = (tt < T && P.observed[tt]) ? 1 : 0;
      double F_l = 1.0;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const int cn = SEAS ? cursor_prev(c, S.ns) : 0;   // the next layout's cursor
        const int rc = S.s0 + c, rw = S.s0 + cn;          // rows / columns of the current and the new first component
        // PZ_k = P(k, 0) + P(k, first seasonal) = P(0, k) + P(first seasonal, k)
        double PZ = 0.0;
        if (mylane) {
          PZ = s_P[lane];
          if (SEAS) PZ += s_P[rc * SSM_MAX + lane];
        }
        const double F = zdot<SEAS>(S, PZ, c) + H;
        if (!(F > 0.0)) { status = CHAIN_FORECAST_VARIANCE; break; }
        const double TPZ = vecT<TREND, SEAS>(S, PZ, lane, c);
        const double K = obs ? TPZ / F : 0.0;
        if (mylane) {
          blk[s * m + lane] = K;
          s_tv[lane] = TPZ;
        }
        if (lane == s) F_l = F;
        // T P T' -- the trend block: row 0 += row 1, then column 0 += column 1
        if (TREND == 2) {
          if (mylane) s_P[lane] = s_P[lane] + s_P[SSM_MAX + lane];
          __builtin_amdgcn_wave_barrier();
          if (mylane) s_P[lane * SSM_MAX] = s_P[lane * SSM_MAX] + s_P[lane * SSM_MAX + 1];
          __builtin_amdgcn_wave_barrier();
        }
        // -- the seasonal block: the row / column of the component that drops out
        // becomes that of the new first component, -sum over the block
        if (SEAS) {
          double cs = 0.0;
          if (mylane) {
#pragma unroll
            for (int q = 0; q < SSM_MAX - 1; ++q)
              if (q < S.ns) cs -= s_P[(TREND + q) * SSM_MAX + lane];
          }
          const double tot = row_total(S.seasonal(lane) ? cs : 0.0);
          __builtin_amdgcn_wave_barrier();
          if (mylane) {
            s_P[rw * SSM_MAX + lane] = cs;
            s_P[lane * SSM_MAX + rw] = cs;
          }
          __builtin_amdgcn_wave_barrier();
          if (lane == rw) s_P[rw * (SSM_MAX + 1)] = -tot;
          __builtin_amdgcn_wave_barrier();
        }
        // - TPZ K' at an observed step (as (TPZ_i TPZ_j) / F: exactly symmetric)
        if (obs) {
          const double Finv = 1.0 / F;
#pragma unroll
          for (int e4 = 0; e4 < SSM_MAX * SSM_MAX / WAVE; ++e4) {
            const int e = lane + WAVE * e4;
            const int i = e >> 4, k2 = e & 15;
            if (i < m && k2 < m) s_P[e] -= (s_tv[i] * s_tv[k2]) * Finv;
          }
          __builtin_amdgcn_wave_barrier();
        }
        // + RQR
        if (lane == 0) s_P[0] += sig2[0];
        if (TREND == 2 && lane == 1) s_P[SSM_MAX + 1] += sig2[1];
        if (SEAS && lane == rw) s_P[rw * (SSM_MAX + 1)] += sig2[2];
        __builtin_amdgcn_wave_barrier();
        c = cn;
      }
      if (status != CHAIN_OK) break;
      blk_store(gK + (size_t)tb * m, blk, nstep * m, lane);
      if (tt < T) sres[tt] = F_l;
    }
    if (status != CHAIN_OK && lane == 0) s_flag = status;
  } else {
    double alpha = 0.0;
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const bool in_l = tt < T;
      const double ys_l = in_l ? w0[tt] : 0.0;
      const int nb_l = (tt == 0) ? 0 : nfirst + (tt - 1) * nper;
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
      double w_l = 0.0;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        if (tb + s == 0) {
          // simulate_initial_state: mean_i + sd_i z_i
          double z = 0.0;
          if (mylane) {
            if (lane < TREND) z = (lane < d0) ? szz[lane] : 0.0;
            else z = szz[d0 + (lane - S.s0)];
          }
          alpha = mylane ? sqrt(P0l) * z + a0l : 0.0;
        } else {
          // simulate_next_state: T alpha + eta
          const double z0 = rl(z0_l, s);
          const int cn = SEAS ? cursor_prev(c, S.ns) : 0;
          alpha = vecT<TREND, SEAS>(S, alpha, lane, c);
          if (TREND == 2) alpha += sd_tr * ((lane == 0) ? z0 : rl(z1_l, s));
          else alpha += sd_tr * z0;
          if (SEAS) { if (lane == S.s0 + cn) alpha += sdv[2] * rl(zs_l, s); }
          c = cn;
        }
        const double yplus = zdot<SEAS>(S, alpha, c) + sqrtH * rl(zh_l, s);   // simulate_adjusted_observation
        const double w = rl(ys_l, s) - yplus;
        if (lane == s) w_l = w;
        if (mylane) blk[s * m + lane] = alpha;
      }
      blk_store(gst + (size_t)tb * m, blk, nstep * m, lane);
      if (in_l) w0[tt] = w_l;
    }
  }
  SSTAMP(3);
  __threadfence_block();
  __syncthreads();
  SSTAMP(4);
  status = s_flag;
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }
  if (wave != 0) return;

  // ---- 3b. the filter on w = y* - y+ (the data filter minus the simulation
  // filter; they share the gains): v - v+ = w - Z'(a - a+); a - a+ <- T (a - a+) + K (v - v+)
  {
    double delta = 0.0;
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const bool in_l = tt < T;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
      blk_load(blk, gK + (size_t)tb * m, nstep * m, lane);
      const double w_l = in_l ? w0[tt] : 0.0, F_l = in_l ? sres[tt] : 1.0;
      const int ob_l = (in_l && P.observed[tt]) ? 1 : 0;
      double ef_l = 0.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double K = mylane ? blk[s * m + lane] : 0.0;
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const double e = obs ? rl(w_l, s) - zdot<SEAS>(S, delta, c) : 0.0;
        if (lane == s) ef_l = obs ? e / F_l : 0.0;
        delta = vecT<TREND, SEAS>(S, delta, lane, c) + K * e;
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
  double r = 0.0;
  for (int tb = ((T - 1) / WAVE) * WAVE; tb >= 0; tb -= WAVE) {
    const int tt = tb + lane;
    const bool in_l = tt < T;
    const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
    blk_load(blk, gK + (size_t)tb * m, nstep * m, lane);
    const double ef_l = in_l ? w0[tt] : 0.0;
    double d0_l = 0.0, d1_l = 0.0, d2_l = 0.0;
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
      const double kr = row_total(K * r);
      const double coef = rl(ef_l, s) - kr;
      r = vecTt<TREND, SEAS>(S, r, lane, c1);
      if (lane == 0 || (SEAS && lane == S.s0 + c0)) r += coef;
      if (!mylane) r = 0.0;
      c1 = c0;
    }
    __builtin_amdgcn_wave_barrier();
    if (in_l) {
      gd[tt] = d0_l;
      if (TREND == 2) gd[(size_t)T + tt] = d1_l;
      if (SEAS) gd[(size_t)2 * T + tt] = d2_l;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  SSTAMP(6);
  // ---- 5. forward: the mean correction E(alpha | y) - E(alpha | y+), the state
  // draw, the state models' and the regression's sufficient statistics
  double mc = P0l * r;          // a0 + P0 r0 - (a0 + P0 r0+)
  double prev = 0.0;            // state_{t-1} (its own layout)
  double suf0 = 0.0, suf2 = 0.0;
  double mv_ybar = 0.0, mv_sumsq = 0.0, mv_n = 0.0;   // MvnSuf of the trend errors (lanes 0, 1)
  double yty = 0.0, nobs = 0.0;
  double *oblk = s_blk[1];
  {
    int c = 0;
    for (int tb = 0; tb < T; tb += WAVE) {
      const int tt = tb + lane;
      const bool in_l = tt < T;
      const int nstep = (T - tb < WAVE) ? T - tb : WAVE;
      blk_load(blk, gst + (size_t)tb * m, nstep * m, lane);
      const bool dd = in_l && tt > 0;
      const double d0_l = dd ? gd[tt - 1] : 0.0;
      const double d1_l = (dd && TREND == 2) ? gd[(size_t)T + tt - 1] : 0.0;
      const double d2_l = (dd && SEAS) ? gd[(size_t)2 * T + tt - 1] : 0.0;
      const double y_l = in_l ? P.y[tt] : 0.0;
      const int ob_l = (in_l && P.observed[tt]) ? 1 : 0;
      double res_l = 0.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double ap = mylane ? blk[s * m + lane] : 0.0;
        if (tb + s > 0) {
          const int cn = SEAS ? cursor_prev(c, S.ns) : 0;
          mc = vecT<TREND, SEAS>(S, mc, lane, c);
          if (TREND == 2) mc += sig_tr * ((lane == 0) ? rl(d0_l, s) : rl(d1_l, s));
          else mc += sig_tr * rl(d0_l, s);
          if (SEAS) { if (lane == S.s0 + cn) mc += sig2[2] * rl(d2_l, s); }
          c = cn;
        }
        const double st = mylane ? ap + mc : 0.0;
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
        }
        prev = st;
        if (mylane) {
          // the state draw goes out in logical order
          int idx = lane;
          if (SEAS && lane >= S.s0) {
            const int q = lane - S.s0;
            idx = S.s0 + (q >= c ? q - c : q - c + S.ns);
          }
          oblk[s * m + idx] = st;
        }
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const double resid = obs ? rl(y_l, s) - zdot<SEAS>(S, st, c) : 0.0;
        if (lane == s) res_l = resid;
        if (obs) { yty += resid * resid; nobs += 1.0; }
      }
      blk_store(gst + (size_t)tb * m, oblk, nstep * m, lane);
      if (in_l) sres[tt] = res_l;
    }
  }
  SSTAMP(7);
#ifdef BA_KSTAMPS
  if (chain == 0 && lane == 0 && draw_variances)
    printf("ssm phases (cycles, wave 0): variances %lld ystar %lld normals %lld sim %lld wait-for-P %lld filter %lld backward %lld correction %lld\n",
           kph[0], kph[1], kph[2], kph[3], kph[4], kph[5], kph[6], kph[7]);
#endif
  // publish the sufficient statistics
  if (TREND == 2) {
    // center_sumsq(mu = 0)(i, i) = sumsq_ii + n ybar_i^2
    const double ssv = mv_sumsq + mv_ybar * mv_ybar * mv_n;
    if (lane < 2) {
      Q.var_n[(size_t)chain * 3 + lane] = mv_n;
      Q.var_ss[(size_t)chain * 3 + lane] = ssv;
    }
  } else if (lane == 0) {
    Q.var_n[(size_t)chain * 3 + 0] = (double)(T - 1);
    Q.var_ss[(size_t)chain * 3 + 0] = suf0;
  }
  if (SEAS) {
    // (the lane that accumulated moved with the cursor: sum over the block)
    const double tot = row_total(S.seasonal(lane) ? suf2 : 0.0);
    if (lane == 0) {
      Q.var_n[(size_t)chain * 3 + 2] = (double)(T - 1);
      Q.var_ss[(size_t)chain * 3 + 2] = tot;
    }
  }
  if (lane == 0) {
    P.yty[chain] = yty;
    P.nobs[chain] = nobs;
    P.status[chain] = status;
  }
}

// StateSpaceRegressionModel::simulate_forecast for every chain's current draw of the
// structural model (StateSpaceRegressionModel.cpp:216-219, :256-278): the state
// advances by T state + state errors (trend, then seasonal), the observation is
// rnorm(Z'state, sigma_obs) + x'beta; normals in the reference's order on the
// chain's forecast stream (id 5).  One wavefront per chain, lane = state component
// (logical order; the horizon is short, the seasonal block simply shifts).
__global__ __launch_bounds__(64) void ssm_forecast_kernel(SsParams P, int horizon, const double *newX,
                                                          uint64_t *pos_forecast, double *out) {
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) return;
  const SsmParams &Q = P.ssm;
  const int T = P.T, p = P.p, m = Q.m, trend = Q.trend, s0 = Q.s0;
  const int ns = Q.nseasons > 0 ? Q.nseasons - 1 : 0;
  const double *beta = P.beta + (size_t)chain * p;
  const double sd_obs = sqrt(P.sigsq[chain]);
  const double sd0 = sqrt(Q.var_sigsq[(size_t)chain * 3 + 0]), sd1 = sqrt(Q.var_sigsq[(size_t)chain * 3 + 1]),
               sd2 = sqrt(Q.var_sigsq[(size_t)chain * 3 + 2]);
  const double *gst = Q.work + (size_t)chain * Q.work_stride + (size_t)m * T;
  double st = (lane < m) ? gst[(size_t)(T - 1) * m + lane] : 0.0;
  const bool seas = ns > 0 && lane >= s0 && lane < s0 + ns;
  SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 5u}, pos_forecast[chain]};
  for (int i = 0; i < horizon; ++i) {
    // state errors, in the reference's order
    double e0, e1 = 0.0, e2 = 0.0;
    if (trend == 1) {
      e0 = d_rnorm(rng, 0.0, sd0);
    } else {
      const double z0 = d_rnorm(rng, 0.0, 1.0), z1 = d_rnorm(rng, 0.0, 1.0);
      e0 = sd0 * z0 + 0.0;
      e1 = sd1 * z1 + 0.0;
    }
    if (ns > 0) e2 = d_rnorm(rng, 0.0, sd2);
    // T state
    double nx = st;
    if (trend == 2) { const double x1 = rl(st, 1); if (lane == 0) nx = st + x1; }
    if (ns > 0) {
      // (first = 0 - s_0 - s_1 - ..., SeasonalStateSpaceMatrix::multiply)
      double first = 0.0;
      for (int q = 0; q < ns; ++q) first -= rl(st, s0 + q);
      const double prev = sdpp<0x111, 0xf>(st, 0.0);
      if (lane == s0) nx = first; else if (seas) nx = prev;
    }
    st = nx + ((lane == 0) ? e0 : ((trend == 2 && lane == 1) ? e1 : ((ns > 0 && lane == s0) ? e2 : 0.0)));
    if (lane >= m) st = 0.0;
    double zs = rl(st, 0);
    if (ns > 0) zs += rl(st, s0);
    const double obs = d_rnorm(rng, zs, sd_obs);
    double part = 0.0;
    for (int j = lane; j < p; j += WAVE) part += newX[(size_t)j * horizon + i] * beta[j];
    // (sum over the wave: 64 lanes)
    part += sdpp<0x111, 0xf>(part, 0.0);
    part += sdpp<0x112, 0xf>(part, 0.0);
    part += sdpp<0x114, 0xf>(part, 0.0);
    part += sdpp<0x118, 0xf>(part, 0.0);
    const double pred = rl(part, 15) + rl(part, 31) + rl(part, 47) + rl(part, 63);
    if (lane == 0) out[(size_t)chain * horizon + i] = obs + pred;
  }
  if (lane == 0) pos_forecast[chain] = rng.pos;
}

hipError_t launch_ssm_forecast(hipStream_t stream, const SsParams &P, int horizon, const double *newX,
                               uint64_t *pos_forecast, double *out) {
  hipLaunchKernelGGL(ssm_forecast_kernel, dim3(P.chain_count), dim3(WAVE), 0, stream, P, horizon, newX,
                     pos_forecast, out);
  return hipGetLastError();
}

hipError_t launch_xte_tiled(hipStream_t stream, const double *U, int64_t ldu, int R, const double *B, int64_t n,
                            int p, double *out, double *planes);

hipError_t launch_ssm_simsmooth(hipStream_t stream, const SsParams &P, int draw_variances) {
  const dim3 grid(P.chain_count), block(2 * WAVE);
  const bool seas = P.ssm.nseasons > 0;
#define SSM_LAUNCH(TR, SE) hipLaunchKernelGGL((ssm_simsmooth_kernel<TR, SE>), grid, block, 0, stream, P, draw_variances)
  hipError_t err;
  {
    KtScope kt(stream, KT_SSM);
    if (!seas) {
      if (P.ssm.trend == 1) SSM_LAUNCH(1, false); else SSM_LAUNCH(2, false);
    } else {
      if (P.ssm.trend == 1) SSM_LAUNCH(1, true); else SSM_LAUNCH(2, true);
    }
    err = hipGetLastError();
  }
#undef SSM_LAUNCH
  if (err != hipSuccess) return err;
  // xty[chain, j] = x_j' e_chain (the residual series are array 1 of every chain's scratch block)
  return launch_xte_tiled(stream, P.scratch + (size_t)P.chain_first * P.scratch_stride + P.T, P.scratch_stride,
                          P.chain_count, P.X, (int64_t)P.T, P.p, P.xty + (size_t)P.chain_first * P.p,
                          P.xte_planes);
}

}  // namespace boom_amd
