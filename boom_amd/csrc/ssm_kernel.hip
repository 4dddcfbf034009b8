// bsts structural time series for many chains: the state half of
// StateSpacePosteriorSampler::draw() for BOOM's block-diagonal state -- ANY list of state
// models in the order they were added with add_state (StateSpaceModelBase.hpp:637-638,
// Filters/SparseMatrix.hpp:2196 BlockDiagonalMatrix):
//   LocalLevelStateModel                     1 component,  ZeroMeanGaussianConjSampler
//   LocalLinearTrendStateModel               2 components, one ZeroMeanMvnIndependenceSampler
//                                            per variance (as bsts builds it)
//   SeasonalStateModel(nseasons, duration)   nseasons - 1 components; T / RQR are the seasonal
//                                            matrices on the steps INTO a new season and
//                                            identity / zero inside one
//                                            (SeasonalStateModel.cpp:89-104, :248-258)
//   ArStateModel(lags)                       lags components, ArPosteriorSampler
//   StaticInterceptStateModel                1 component, no state error, no sampler: here a
//                                            local level whose variance slot stays 0
//                                            (StaticInterceptStateModel.hpp:35-131)
//   TrigStateModel(period, frequencies)      2 components per frequency that rotate, Z = 1 at
//                                            every pair's first, ONE variance for all of them
//                                            (TrigStateModel.cpp:130-223)
//   SemilocalLinearTrendStateModel           3 components (level, slope, the slope's long-run
//                                            mean mu): T = [[1, 1, 0], [0, phi, 1 - phi], [0, 0, 1]],
//                                            errors on level and slope; the level's
//                                            ZeroMeanGaussianConjSampler and the slope's
//                                            NonzeroMeanAr1Sampler (SemilocalLinearTrend.cpp:29-272,
//                                            NonzeroMeanAr1Sampler.cpp:51-155)
// SURVEY 8f row f2.
//
//   state model samplers                 (ZeroMeanGaussianConjSampler.cpp:57-60,
//                                         ZeroMeanMvnIndependenceSampler.cpp:63-70)
//   Base::impute_state                   (StateSpaceModelBase.cpp:278-291)
//     ScalarBase::simulate_forward       (:771-790) with
//       ScalarMarginalDistribution::update (ScalarKalmanFilter.cpp:41-83), vector state
//       StateModelBase::simulate_initial_state (StateModel.cpp:47-56)
//       simulate_state_error (LocalLevelStateModel.cpp:62-64, MvnBase.cpp:257,
//                             SeasonalStateModel.cpp:124-146, ArStateModel.cpp:85-90)
//     Base::propagate_disturbances       (:858-891), fast_disturbance_smooth
//                                          (ScalarKalmanFilter.cpp:168-196)
//     observe_state (LocalLevelStateModel.cpp:52-58, LocalLinearTrend.cpp:53-63,
//                    SeasonalStateModel.cpp:74-86, ArStateModel.cpp:64-69),
//     observe_data_given_state
//   ArPosteriorSampler::draw             (ArPosteriorSampler.cpp:52-143)
//
// State vector = the blocks one after the other, dimension m <= 64: lane j of a wavefront
// holds component j of every state-sized vector.
//   Z    ones at the first element of each block
//   T    local level [1]; trend [[1, 1], [0, 1]]; seasonal: first row -1, ones below the
//        diagonal; autoregression: first row phi, ones below the diagonal
//   RQR  diagonal: level; level, slope; the first element of a seasonal / autoregression block
// One chain per workgroup of two wavefronts: both share the adjusted observations and the
// sweep's normals (stream_normals.h); then wave 1 runs the variance recursion (P_t, F_t,
// K_t: it does not look at the data) while wave 0 simulates alpha+, y+, and wave 0 goes on
// with the filter on w = y* - y+ (the data filter and the simulation filter share the
// gains, so ONE filter runs on the difference), the backward pass and the mean correction.
// The passes are SERIAL in time (the per-step maps are m x m and their compositions do not
// fit a wave scan), so the design is about the length of a step's dependent chain:
//   * every seasonal block sits in a ROTATING layout (its own cursor, advanced on the
//     steps into a new season only): the transition moves nothing;
//   * the state variance P lives in LDS (leading dimension odd: a lane per column and a
//     lane per row are both conflict-free) and is advanced in the FILTERED form
//     P_{t+1} = T (P_t - PZ PZ' / F) T' + RQR -- the same matrix as the reference's
//     T P T' - (T PZ) K' + RQR -- because then one pass of lane k over ITS column applies
//     the rank-one term and T from the left (three LDS round trips per step: the rows Z
//     selects, the column pass, the pass over the lane's row that applies T' from the
//     right).  (PZ_i PZ_k) / F is formed as a commutative product first and the two passes
//     sum in the same order, so P stays exactly symmetric.
#include <hip/hip_runtime.h>

#include "diag.h"
#include "ktimer.h"

#include "device_rng.h"
#include "kalman_params.h"
#include "stream_normals.h"

#include "ssg_device.h"

namespace boom_amd {

namespace {
constexpr int SSG_V_FAILED = 1 << 30;   // s_vprog: the variance pass stopped (F <= 0)
}  // namespace

// LDS of the passes, in doubles: two block buffers (bl x m each) | P (m x ld) | a block's
// normals / smoothed disturbances | the autoregression blocks' xtx rows
__host__ __device__ inline int ssg_pass_lds_doubles(int m, int ld, int bl, int nerr, int nar) {
  return 2 * bl * m + m * ld + (bl * (nerr + 1) + SSG_MAX_STATE + 8) + nar * AR_MAX * (AR_MAX + 1);
}

// grid = chains, block = 128, dynamic LDS = max(the normals generator's lists, the
// sampler's matrices, ssg_pass_lds_doubles).  SMALL: m <= 16.
// LDC: the leading dimension of P as a compile-time constant (17 / 33 / 61 / 65, chosen from
// the state dimension by ssg_finish): an entry's LDS address is then an immediate offset from
// the lane's column or row, where a run-time ld cost an address computation per entry.
// GLOB: the list holds a trig or a semilocal-linear-trend block (round 6): their per-step code
// (pair rotations, the 3 x 3 trend block, the symmetrisation of their rows of P) is compiled into
// the GLOB = true instances only -- carried by every list it cost the round-4 lists 6 - 12 %.
template <bool SMALL, int LDC, bool GLOB>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) void ssg_simsmooth_kernel(SsParams P, int draw_variances) {
  constexpr int SSG_BATCH = SMALL ? 4 : 8;   // entries of a column / row of P asked of the LDS together
  extern __shared__ __align__(16) unsigned char s_raw[];
  __shared__ int s_flag;
  __shared__ int s_vprog;            // blocks the variance pass has put out (wave 1 -> wave 0)
  __shared__ int s_cprog, s_cdone;   // the last pass: blocks of state draws wave 0 has made / wave 1 has taken
  __shared__ double s_sig2[SSG_MAX_VAR];
  __shared__ double s_phi[SSG_MAX_AR * AR_MAX];
  __shared__ double s_tv[SSG_MAX_STATE];
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) return;
  if (P.only_ran && P.only_ran[chain] == 0) return;
  const SsmParams &M = P.ssm;
  const SsgSpec &Q = *M.spec;
  const int T = P.T, p = P.p, m = M.m, nb = M.nblocks, BL = M.bl, NE = M.nerr;
  constexpr int ld = LDC;   // (== M.ld: launch_ssm_simsmooth picks the instance by it)
  NormalsLds &s_norm = *reinterpret_cast<NormalsLds *>(s_raw);
  ArLds &s_ar = *reinterpret_cast<ArLds *>(s_raw);
  double *s_blk0 = reinterpret_cast<double *>(s_raw);
  double *s_blk1 = s_blk0 + BL * m;
  double *s_P = s_blk1 + BL * m;
  double *s_z = s_P + m * ld;
  double *s_axx = s_z + (BL * (NE + 1) + SSG_MAX_STATE + 8);
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  int status = CHAIN_OK;
  if (threadIdx.x == 0) { s_flag = CHAIN_OK; s_vprog = 0; s_cprog = 0; s_cdone = 0; }
  if (threadIdx.x < SSG_MAX_VAR) s_sig2[threadIdx.x] = M.var_sigsq[(size_t)chain * SSG_MAX_VAR + threadIdx.x];
  __syncthreads();
#ifdef BA_KSTAMPS
  long long kph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, klast = (long long)__builtin_readcyclecounter();
#endif

  // ---- the state models' samplers, in model order (every sampler reads its own stream)
  if (draw_variances) {
    for (int b = 0; b < nb; ++b) {
      const SsgBlock &K = Q.blk[b];
      if (K.kind == SSG_AR) continue;
      // (a semilocal trend: the level's variance here; the slope's NonzeroMeanAr1Sampler below, after it)
      for (int v = 0; v < (K.kind == SSG_SEMILOCAL ? 1 : K.nvar); ++v) {
        const int vi = K.var0 + v;
        const size_t at = (size_t)chain * SSG_MAX_VAR + vi;
        SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, (uint32_t)K.sid[v]}, M.pos_var[at]};
        int bad = 0;
        const double DF = M.var_n[at] + Q.prior_df[vi];
        const double SSQ = M.var_ss[at] + Q.prior_ss[vi];
        double draw = d_draw_variance(rng, DF, SSQ, Q.sigma_max[vi], &bad);
        if (bad) status = CHAIN_RNG_BRANCH;
        // ZeroMeanMvnIndependenceSampler sets siginv(i, i) = 1 / draw; the model's
        // Sigma is the inverse of that again
        if (K.kind == SSG_LOCAL_LINEAR_TREND) draw = 1.0 / (1.0 / draw);
        __syncthreads();   // (everybody has read the old position)
        if (threadIdx.x == 0) {
          M.pos_var[at] = rng.pos;
          M.var_sigsq[at] = draw;
          s_sig2[vi] = draw;
        }
      }
    }
  }
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }
  // ---- the autoregression blocks' samplers, by wave 0 (the sampler's vectors sit at lanes 0 .. L - 1)
  for (int b = 0; b < nb; ++b) {
    const SsgBlock &K = Q.blk[b];
    if (K.kind == SSG_SEMILOCAL) {
      // the slope model's sampler (mu, phi, sigma: one stream), by wave 0, every lane alike
      const int vi = K.var0 + 1;
      const size_t at = (size_t)chain * SSG_MAX_VAR + vi;
      double *gphi = M.ar_phi + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_MAX;
      if (wave == 0) {
        double ph = gphi[0], mu = gphi[1], sig2s = M.var_sigsq[at];
        if (draw_variances) {
          SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, (uint32_t)K.sid[1]}, M.pos_var[at]};
          const double *suf = M.ar_suf + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_SUF_STRIDE;
          const int st = semilocal_slope_draw(suf, Q.sl_prior[K.ar_index], K.sl_truncate != 0, K.sl_positive != 0,
                                              Q.prior_df[vi], Q.prior_ss[vi], Q.sigma_max[vi], rng, mu, ph, sig2s);
          if (st != CHAIN_OK) {
            if (lane == 0) s_flag = st;
          } else if (lane == 0) {
            gphi[0] = ph;
            gphi[1] = mu;
            M.var_sigsq[at] = sig2s;
            M.pos_var[at] = rng.pos;
          }
        }
        if (lane < AR_MAX) s_phi[K.ar_index * AR_MAX + lane] = lane == 0 ? ph : (lane == 1 ? mu : 0.0);
        if (lane == 0) s_sig2[vi] = sig2s;
      }
      __syncthreads();
      continue;
    }
    if (K.kind != SSG_AR) continue;
    const int L = K.lags, vi = K.var0;
    const size_t at = (size_t)chain * SSG_MAX_VAR + vi;
    double *gphi = M.ar_phi + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_MAX;
    if (wave == 0) {
      double ph = (lane < L) ? gphi[lane] : 0.0;
      double sig2a = M.var_sigsq[at];
      if (draw_variances) {
        SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, (uint32_t)K.sid[0]}, M.pos_var[at]};
        const double *suf = M.ar_suf + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_SUF_STRIDE;
        const int st = ar_draw(s_ar, suf, L, Q.prior_df[vi], Q.prior_ss[vi], Q.sigma_max[vi], rng, ph, sig2a, lane);
        if (st != CHAIN_OK) {
          if (lane == 0) s_flag = st;
        } else {
          if (lane < L) gphi[lane] = ph;
          if (lane == 0) {
            M.var_sigsq[at] = sig2a;
            M.pos_var[at] = rng.pos;
          }
        }
      }
      if (lane < AR_MAX) s_phi[K.ar_index * AR_MAX + lane] = (lane < L) ? ph : 0.0;
      if (lane == 0) s_sig2[vi] = sig2a;
    }
    __syncthreads();
  }
  __syncthreads();
  status = s_flag;
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }

  // ---- the block list, the lane's own constants
  Blocks B;
  B.load(Q, nb, lane);
  LaneInfo LI{-1, 0, 0, 0, 0, 0.0};
  int var_l = 0;          // the variance parameter behind this lane's state error
  int erow_l = 0;         // which of the state-error rows this lane's is (the smoothed disturbances' series)
  bool sl_mean_lane = false;   // the lane holds a semilocal trend's third component (the slope's long-run mean)
  double sl_mu_l = 0.0;        // ... and its value
  int cbefore_l = 0;      // error terms drawn at EVERY step ahead of this lane's term
  unsigned sbefore_l = 0; // seasonal blocks ahead of it (their terms are drawn on some steps only)
  int ipos_l = 0;         // position of this lane's normal among those of the initial state
  bool init_l = false;    // ... if it has one
  double a0l = 0.0, P0l = 0.0;
  int nconst_err = 0;     // error normals drawn at every step
  unsigned seas_active = 0;   // seasonal blocks whose error is drawn at all (sigma != 0)
  int nfirst = 0;         // normals of the initial state
  {
    int cb = 0, ip = 0, eb = 0;
    unsigned sb = 0;
    for (int b = 0; b < nb; ++b) {
      const unsigned d = B.udesc(b);
      const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d), v0 = Blocks::var0_of(d);
      const bool mine = lane >= f && lane < f + n;
      const bool second = (kd == SSG_LOCAL_LINEAR_TREND || kd == SSG_SEMILOCAL) && lane == f + 1;
      const int within = kd == SSG_TRIG ? lane - f : (second ? 1 : 0);   // (a trig block: an error term per component)
      if (mine) {
        LI.blk = b; LI.kind = kd; LI.first = f; LI.dim = n;
        var_l = v0 + (second ? 1 : 0);
        cbefore_l = cb + within;
        erow_l = eb + within;
        sbefore_l = sb;
        if (kd == SSG_AR) LI.phi = s_phi[Blocks::arx_of(d) * AR_MAX + (lane - f)];
        if (kd == SSG_TRIG) { LI.tc = Q.trig_c[lane]; LI.ts = Q.trig_s[lane]; }
        if (kd == SSG_SEMILOCAL) {
          LI.phi = s_phi[Blocks::arx_of(d) * AR_MAX];
          // (the third component, the slope's long-run mean: no error term -- its variance slot is
          // the slope's, switched off below -- and its initial mean is mu as it stands)
          if (lane == f + 2) { var_l = -1; sl_mu_l = s_phi[Blocks::arx_of(d) * AR_MAX + 1]; sl_mean_lane = true; }
        }
      }
      eb += kd == SSG_TRIG ? n : ((kd == SSG_LOCAL_LINEAR_TREND || kd == SSG_SEMILOCAL) ? 2 : 1);
      // the initial state's normals: a local level draws rnorm(a0, sd0) (nothing when
      // sd0 == 0), every other model rmvn: one per component
      if (kd == SSG_LOCAL_LEVEL) {
        const bool drawn = Q.P0[f] != 0.0;
        if (mine) { ipos_l = ip; init_l = drawn; }
        ip += drawn ? 1 : 0;
      } else if (kd == SSG_SEMILOCAL) {
        // rnorm_mt(level mean, sd), rnorm_mt(slope mean, sd), mu (SemilocalLinearTrend.cpp:262-270)
        if (mine) { ipos_l = ip + (lane - f); init_l = lane < f + 2; }
        ip += 2;
      } else {
        if (mine) { ipos_l = ip + (lane - f); init_l = true; }
        ip += n;
      }
      // the state errors of a step: local level: one if sigma != 0; trend: two, always;
      // seasonal: one on the steps into a new season if sigma != 0; autoregression: one, always;
      // trig: rnorm_mt(0, sigma) per component, i.e. dim of them if sigma != 0
      const bool nz = s_sig2[v0] != 0.0;
      if (kd == SSG_LOCAL_LEVEL) cb += nz ? 1 : 0;
      else if (kd == SSG_LOCAL_LINEAR_TREND) cb += 2;
      else if (kd == SSG_SEMILOCAL) cb += 2;   // (rnorm_mt(0, sigma) for level and slope: both sigmas are positive)
      else if (kd == SSG_AR) cb += 1;
      else if (kd == SSG_TRIG) cb += nz ? n : 0;
      else {
        if (nz) seas_active |= 1u << b;
        sb |= 1u << b;
      }
    }
    nconst_err = __builtin_amdgcn_readfirstlane(cb);
    nfirst = __builtin_amdgcn_readfirstlane(ip);
    seas_active = (unsigned)__builtin_amdgcn_readfirstlane((int)seas_active);
  }
  const bool mylane = lane < m;
  if (mylane) { a0l = sl_mean_lane ? sl_mu_l : Q.a0[lane]; P0l = Q.P0[lane]; }
  const double sig_l = (mylane && var_l >= 0) ? s_sig2[var_l] : 0.0;
  const double sd_l = sqrt(sig_l);

  const double H = P.sigsq[chain], sqrtH = sqrt(H);
  const int dH = (sqrtH != 0.0);
  const double *beta = P.beta + (size_t)chain * p;
  double *w0 = P.scratch + (size_t)chain * P.scratch_stride;   // y* -> w = y* - y+ -> (v - v+) / F
  double *sres = w0 + T;                                       // F_t, then residuals (input of the X'e GEMM)
  double *wk = M.work + (size_t)chain * M.work_stride;
  double *gK = wk;                                 // K_t, m per step (layout of step t + 1)
  double *gst = gK + (size_t)m * T;                // alpha+_t (layout of step t), then the state draw
  double *gd = gst + (size_t)m * T;                // r_t (difference) at the rows with state error: nerr series of T
  double *szz = gd + (size_t)NE * T;               // the sweep's normals

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
        const double bb = rl(bj, l);
        pred += P.X[(size_t)(base + l) * T + (t < T ? t : T - 1)] * bb;
      }
    }
    if (t < T) w0[t] = P.y[t] - pred;
  }

  SSTAMP(1);
  // ---- 2. the normals of simulate_forward, in stream order.  t = 0: the initial state of
  // every state model, then the observation; t >= 1: the state errors of the step into t
  // (model by model), then the observation.  zoffset(t) = index of step t's first normal.
  auto seasonal_draws = [&](int t) -> int {   // seasonal error draws over the steps into times 1 .. t
    int o = 0;
    unsigned sm = seas_active;
    while (sm) {
      const int b = __ffs((int)sm) - 1;
      sm &= sm - 1;
      const unsigned dpw = (unsigned)__builtin_amdgcn_readlane((int)B.dp, b);
      o += seasons_started(t, (int)(dpw & 0xffffu), (int)(dpw >> 16));
    }
    return o;
  };
  const int N = (nfirst + dH) + (T - 1) * (nconst_err + dH) + seasonal_draws(T - 1);
  status = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, P.pos_state[chain], N,
                          szz, &P.pos_state[chain], ss_slot_serve(P));
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }
  __syncthreads();
  SSTAMP(2);
  auto zoffset = [&](int t) -> int {   // t >= 1
    return (nfirst + dH) + (t - 1) * (nconst_err + dH) + seasonal_draws(t - 1);
  };

  double *blk = wave == 0 ? s_blk0 : s_blk1;

  // ---- 3. forward, the two waves side by side (neither needs the other's results):
  //   wave 1: the variances P_t -> F_t, K_t (ScalarMarginalDistribution::update, the
  //           part that does not look at the data);
  //   wave 0: simulate alpha+_t, y+_t and w_t = y*_t - y+_t.
  // Time runs in blocks of BL steps: a block's scalar inputs sit one step per lane
  // (read with v_readlane), its state-sized series in LDS, and what a block produces
  // goes out in one coalesced piece.
  if (wave == 1) {
    for (int e = lane; e < m * ld; e += WAVE) s_P[e] = 0.0;
    wave_lds_sync();
    if (mylane) s_P[lane * ld + lane] = P0l;
    wave_lds_sync();
    seek(B, LI, 0, 0);
    for (int tb = 0; tb < T; tb += BL) {
      const int tt = tb + lane;
      const int nstep = (T - tb < BL) ? T - tb : BL;
      const int ob_l = (lane < nstep && P.observed[tt]) ? 1 : 0;
      double F_l = 1.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const unsigned mv = B.moving();
        // PZ_k = sum over the blocks of P(first of the block, k)  [= P(k, first), P symmetric]
        // (the blocks' rows are asked for together: one LDS round trip, not one per block)
        double PZ = 0.0;
        if (SMALL) {
#pragma nounroll
          for (int b = 0; b < nb; ++b) {
            const int zl = Blocks::first_of(B.udesc(b)) + (int)(B.urc(b) >> 16);
            if (mylane) PZ += s_P[zl * ld + lane];
          }
        } else {
          double pz[SSG_MAX_BLOCKS];
#pragma unroll
          for (int b = 0; b < SSG_MAX_BLOCKS; ++b) {
            const int zl = Blocks::first_of(B.udesc(b)) + (int)(B.urc(b) >> 16);
            pz[b] = (b < nb && mylane) ? s_P[zl * ld + lane] : 0.0;
          }
#pragma unroll
          for (int b = 0; b < SSG_MAX_BLOCKS; ++b)
            if (b < nb) PZ += pz[b];
        }
        // (a trig block: Z selects every pair's first component, not the block's alone)
        if (GLOB) {
          unsigned tq = B.trigmask;
          while (tq) {
            const int b = __ffs((int)tq) - 1;
            tq &= tq - 1;
            const unsigned d = B.udesc(b);
            const int f = Blocks::first_of(d), n = Blocks::dim_of(d);
#pragma nounroll
            for (int i = 2; i < n; i += 2)
              if (mylane) PZ += s_P[(f + i) * ld + lane];
          }
        }
        const double F = zdot<SMALL, GLOB>(LI, PZ, lane) + H;
        if (!(F > 0.0)) { status = CHAIN_FORECAST_VARIANCE; break; }
        if (lane == s) F_l = F;
        const double Finv = 1.0 / F;
        // K_t = T PZ / F (layout of t + 1)
        const double TPZ = vecT<SMALL, GLOB>(B, LI, PZ, lane, mv);
        if (mylane) {
          blk[s * m + lane] = obs ? TPZ * Finv : 0.0;
          s_tv[lane] = PZ;
        }
        wave_lds_sync();
        // -- the column pass: lane k walks ITS column: the rank-one term of an observed step,
        // then T from the left, block by block
#pragma nounroll
        for (int b = 0; b < nb; ++b) {
          const unsigned d = B.udesc(b);
          const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d);
          if (!mylane) continue;
          double *col = s_P + f * ld + lane;
          if (kd == SSG_LOCAL_LEVEL) {
            double v = col[0];
            if (obs) v -= (s_tv[f] * PZ) * Finv;
            if (lane == f) v += s_sig2[Blocks::var0_of(d)];   // (+ RQR: this block's T is the identity)
            col[0] = v;
          } else if (kd == SSG_LOCAL_LINEAR_TREND) {
            double v0 = col[0], v1 = col[ld];
            if (obs) {
              v0 -= (s_tv[f] * PZ) * Finv;
              v1 -= (s_tv[f + 1] * PZ) * Finv;
              col[ld] = v1;
            }
            col[0] = v0 + v1;   // row 0 += row 1
          } else if (kd == SSG_SEASONAL) {
            const bool moves = (mv >> b) & 1u;
            if (obs || moves) {
              // (eight entries of the column in flight at a time: a rolled walk waited for
              // one LDS round trip per entry -- 51 of them a step in bsts's daily model)
              double cs = 0.0;
              int i0 = 0;
              {
                // whole batches: no guards, the addresses immediates off the column's start
#pragma nounroll
                for (; i0 + SSG_BATCH <= n; i0 += SSG_BATCH) {
                  double *c0 = col + i0 * ld;
                  const double *t0 = s_tv + f + i0;
                  double v[SSG_BATCH], tv[SSG_BATCH];
#pragma unroll
                  for (int u = 0; u < SSG_BATCH; ++u) { v[u] = c0[u * ld]; tv[u] = t0[u]; }
                  if (obs) {
#pragma unroll
                    for (int u = 0; u < SSG_BATCH; ++u) {
                      v[u] -= (tv[u] * PZ) * Finv;
                      c0[u * ld] = v[u];
                    }
                  }
#pragma unroll
                  for (int u = 0; u < SSG_BATCH; ++u) cs -= v[u];
                }
              }
#pragma nounroll
              for (; i0 < n; ++i0) {   // (what is left of the block, one by one)
                double v = col[i0 * ld];
                if (obs) {
                  v -= (s_tv[f + i0] * PZ) * Finv;
                  col[i0 * ld] = v;
                }
                cs -= v;
              }
              // the row of the component that drops out becomes that of the new first
              // component, -(sum over the block)
              if (moves) col[sprev((int)(B.urc(b) >> 16), n) * ld] = cs;
            }
          } else if (GLOB && kd == SSG_SEMILOCAL) {
            double v0 = col[0], v1 = col[ld], v2 = col[2 * ld];
            if (obs) {
              v0 -= (s_tv[f] * PZ) * Finv;
              v1 -= (s_tv[f + 1] * PZ) * Finv;
              v2 -= (s_tv[f + 2] * PZ) * Finv;
              col[2 * ld] = v2;
            }
            const double ph = s_phi[Blocks::arx_of(d) * AR_MAX];
            col[0] = v0 + v1;
            col[ld] = ph * v1 + (1 - ph) * v2;
          } else if (GLOB && kd == SSG_TRIG) {
            // the rotations from the left, a pair of the column's entries at a time
#pragma nounroll
            for (int i = 0; i < n; i += 2) {
              double v0 = col[i * ld], v1 = col[(i + 1) * ld];
              if (obs) {
                v0 -= (s_tv[f + i] * PZ) * Finv;
                v1 -= (s_tv[f + i + 1] * PZ) * Finv;
              }
              const double c = Q.trig_c[f + i], sn = Q.trig_s[f + i];
              col[i * ld] = c * v0 + sn * v1;
              col[(i + 1) * ld] = -sn * v0 + c * v1;
            }
          } else {
            // autoregression (logical order): from the last lag down, moving each entry
            // one place on as it is read
            const double *ph = s_phi + Blocks::arx_of(d) * AR_MAX;
            double cs = 0.0;
#pragma nounroll
            for (int q0 = n - 1; q0 >= 0; q0 -= SSG_BATCH) {
              const int nn = q0 + 1;   // entries left, q0 the highest of them
              double v[SSG_BATCH], tv[SSG_BATCH], pc[SSG_BATCH];
#pragma unroll
              for (int u = 0; u < SSG_BATCH; ++u) {
                const int q = q0 - (u < nn ? u : nn - 1);
                v[u] = col[q * ld];
                tv[u] = s_tv[f + q];
                pc[u] = ph[q];
              }
#pragma unroll
              for (int u = 0; u < SSG_BATCH; ++u) {
                if (u < nn) {
                  const int q = q0 - u;
                  if (obs) v[u] -= (tv[u] * PZ) * Finv;
                  cs += pc[u] * v[u];
                  if (q + 1 < n) col[(q + 1) * ld] = v[u];
                }
              }
            }
            col[0] = cs;
          }
        }
        wave_lds_sync();
        // -- the row pass: lane k walks ITS row: T' from the right, + RQR
        unsigned tm = mv & ~(unsigned)__ballot(Blocks::kind_of(B.desc) == SSG_LOCAL_LEVEL);
        while (tm) {
          const int b = __ffs((int)tm) - 1;
          tm &= tm - 1;
          const unsigned d = B.udesc(b);
          const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d);
          if (!mylane) continue;
          double *row = s_P + lane * ld + f;
          const double sg = s_sig2[Blocks::var0_of(d)];
          if (kd == SSG_LOCAL_LINEAR_TREND) {
            const double a = row[0], bb = row[1];
            row[0] = (a + bb) + (lane == f ? sg : 0.0);   // column 0 += column 1
            if (lane == f + 1) row[1] = bb + s_sig2[Blocks::var0_of(d) + 1];
          } else if (kd == SSG_SEASONAL) {
            const int w = sprev((int)(B.urc(b) >> 16), n);
            double cs = 0.0;
            int j0 = 0;
            {
#pragma nounroll
              for (; j0 + SSG_BATCH <= n; j0 += SSG_BATCH) {
                double v[SSG_BATCH];
#pragma unroll
                for (int u = 0; u < SSG_BATCH; ++u) v[u] = row[j0 + u];
#pragma unroll
                for (int u = 0; u < SSG_BATCH; ++u) cs -= v[u];
              }
            }
#pragma nounroll
            for (; j0 < n; ++j0) cs -= row[j0];
            row[w] = cs + (lane == f + w ? sg : 0.0);
          } else if (GLOB && kd == SSG_SEMILOCAL) {
            const double r0 = row[0], r1 = row[1], r2 = row[2];
            const double ph = s_phi[Blocks::arx_of(d) * AR_MAX];
            row[0] = (r0 + r1) + (lane == f ? sg : 0.0);
            row[1] = (ph * r1 + (1 - ph) * r2) + (lane == f + 1 ? s_sig2[Blocks::var0_of(d) + 1] : 0.0);
          } else if (GLOB && kd == SSG_TRIG) {
            // the rotations' transposes from the right, + RQR (sigma^2 on the block's whole diagonal)
#pragma nounroll
            for (int j = 0; j < n; j += 2) {
              const double r0 = row[j], r1 = row[j + 1];
              const double c = Q.trig_c[f + j], sn = Q.trig_s[f + j];
              row[j] = (c * r0 + sn * r1) + (lane == f + j ? sg : 0.0);
              row[j + 1] = (-sn * r0 + c * r1) + (lane == f + j + 1 ? sg : 0.0);
            }
          } else {
            const double *ph = s_phi + Blocks::arx_of(d) * AR_MAX;
            double cs = 0.0;
#pragma nounroll
            for (int q0 = n - 1; q0 >= 0; q0 -= SSG_BATCH) {
              const int nn = q0 + 1;
              double v[SSG_BATCH], pc[SSG_BATCH];
#pragma unroll
              for (int u = 0; u < SSG_BATCH; ++u) {
                const int q = q0 - (u < nn ? u : nn - 1);
                v[u] = row[q];
                pc[u] = ph[q];
              }
#pragma unroll
              for (int u = 0; u < SSG_BATCH; ++u) {
                if (u < nn) {
                  const int q = q0 - u;
                  cs += pc[u] * v[u];
                  if (q + 1 < n) row[q + 1] = v[u];
                }
              }
            }
            row[0] = cs + (lane == f ? sg : 0.0);
          }
        }
        wave_lds_sync();
        // -- a trig block's two passes are not the same sums in the same order (T from the left
        // in one, another block's T' from the right in the other): its rows and columns are made
        // symmetric the way the reference does after every update (fix_near_symmetry,
        // SpdMatrix.cpp:350-357) -- lane k averages P(i, k) and P(k, i) for the block's rows i
        if (GLOB && (B.trigmask | B.slmask)) {
          unsigned tq = B.trigmask | B.slmask;
          while (tq) {
            const int b = __ffs((int)tq) - 1;
            tq &= tq - 1;
            const unsigned d = B.udesc(b);
            const int f = Blocks::first_of(d), n = Blocks::dim_of(d);
#pragma nounroll
            for (int i = 0; i + 1 < n; i += 2) {
              if (!mylane) continue;
              double *cu = s_P + (f + i) * ld + lane, *ro = s_P + lane * ld + f + i;
              const double a0 = cu[0], a1 = cu[ld], b0 = ro[0], b1 = ro[1];
              const double m0 = .5 * (a0 + b0), m1 = .5 * (a1 + b1);
              cu[0] = m0; ro[0] = m0;
              cu[ld] = m1; ro[1] = m1;
            }
            if ((n & 1) && mylane) {   // (a semilocal trend's third row)
              double *cu = s_P + (f + n - 1) * ld + lane, *ro = s_P + lane * ld + f + n - 1;
              const double m0 = .5 * (cu[0] + ro[0]);
              cu[0] = m0; ro[0] = m0;
            }
          }
          wave_lds_sync();
        }
        advance(B, LI, mv, lane);
      }
      if (status != CHAIN_OK) break;
      blk_store(gK + (size_t)tb * m, blk, nstep * m, lane);
      if (lane < nstep) sres[tt] = F_l;
      // the block is out: the filter (wave 0, once it has simulated) follows a block behind
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&s_vprog, tb / BL + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (status != CHAIN_OK) {
      if (lane == 0) {
        s_flag = status;
        __hip_atomic_store(&s_vprog, SSG_V_FAILED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      return;
    }
  } else {
    double alpha = 0.0;
    seek(B, LI, 0, -1);   // (at time t the transition INTO t)
    for (int tb = 0; tb < T; tb += BL) {
      const int tt = tb + lane;
      const int nstep = (T - tb < BL) ? T - tb : BL;
      const bool in_l = lane < nstep;
      const double ys_l = in_l ? w0[tt] : 0.0;
      // the block's normals, in stream order
      const int zstart = tb == 0 ? 0 : zoffset(tb);
      const int zend = (tb + nstep >= T) ? N : zoffset(tb + nstep);
      blk_load(s_z, szz + zstart, zend - zstart, lane);
      int zo = 0;
      double w_l = 0.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        if (tb + s == 0) {
          // simulate_initial_state: mean_i + sd_i z_i
          const double z = (mylane && init_l) ? s_z[ipos_l] : 0.0;
          alpha = mylane ? sqrt(P0l) * z + a0l : 0.0;
          zo = nfirst;
          advance(B, LI, 0u, lane);
        } else {
          // simulate_next_state: T alpha + eta
          const unsigned mv = B.moving();
          const unsigned act = mv & seas_active;
          alpha = vecT<SMALL, GLOB>(B, LI, alpha, lane, mv);
          advance(B, LI, mv, lane);
          bool err = false;
          if (LI.kind == SSG_LOCAL_LEVEL) err = sig_l != 0.0;
          else if (LI.kind == SSG_LOCAL_LINEAR_TREND) err = true;
          else if (LI.kind == SSG_AR) err = lane == LI.first;
          else if (GLOB && LI.kind == SSG_TRIG) err = sig_l != 0.0;
          else if (GLOB && LI.kind == SSG_SEMILOCAL) err = lane < LI.first + 2;
          else if (LI.kind == SSG_SEASONAL) err = ((act >> LI.blk) & 1u) && lane == LI.first + LI.cur;
          const double z = err ? s_z[zo + cbefore_l + __popc(act & sbefore_l)] : 0.0;
          alpha += sd_l * z;
          zo += nconst_err + __popc(act);
        }
        const double zh = dH ? s_z[zo] : 0.0;
        zo += dH;
        const double yplus = zdot<SMALL, GLOB>(LI, alpha, lane) + sqrtH * zh;   // simulate_adjusted_observation
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
    seek(B, LI, 0, 0);
    for (int tb = 0; tb < T; tb += BL) {
      const int tt = tb + lane;
      const int nstep = (T - tb < BL) ? T - tb : BL;
      const bool in_l = lane < nstep;
      // (the gains and F_t of this block: wave 1 is somewhere ahead, or about to be)
      int vp;
      while ((vp = __hip_atomic_load(&s_vprog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) <= tb / BL)
        __builtin_amdgcn_s_sleep(8);
      if (vp == SSG_V_FAILED) {
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
        const unsigned mv = B.moving();
        const double e = obs ? rl(w_l, s) - zdot<SMALL, GLOB>(LI, delta, lane) : 0.0;
        if (lane == s) ef_l = obs ? e / F_l : 0.0;
        delta = vecT<SMALL, GLOB>(B, LI, delta, lane, mv) + K * e;
        advance(B, LI, mv, lane);
      }
      __builtin_amdgcn_wave_barrier();
      if (in_l) w0[tt] = ef_l;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  SSTAMP(5);
  // ---- 4. backward: fast_disturbance_smooth for d = r - r+:
  // r_{t-1} = T_t' r_t + Z ((v_t - v+_t) / F_t - K_t' r_t), r_{T-1} = 0.  r_t is in the
  // layout of step t + 1.  What the correction pass needs of r_t is its value at the rows
  // that carry state error: one series per variance parameter.
  seek(B, LI, T, -1);   // (the layout of time T, the transition T - 1)
  for (int tb = ((T - 1) / BL) * BL; tb >= 0; tb -= BL) {
    const int tt = tb + lane;
    const int nstep = (T - tb < BL) ? T - tb : BL;
    const bool in_l = lane < nstep;
    blk_load(blk, gK + (size_t)tb * m, nstep * m, lane);
    const double ef_l = in_l ? w0[tt] : 0.0;
#pragma nounroll
    for (int s = nstep - 1; s >= 0; --s) {
      const double K = mylane ? blk[s * m + lane] : 0.0;
      const unsigned mv = B.moving();
      // r_t at the error rows (layout of t + 1), one value per step and variance parameter
      if (mylane) {
        bool carrier;
        if (LI.kind == SSG_SEASONAL) carrier = lane == LI.first + LI.cur;
        else if (LI.kind == SSG_LOCAL_LINEAR_TREND || (GLOB && LI.kind == SSG_TRIG)) carrier = true;
        else if (GLOB && LI.kind == SSG_SEMILOCAL) carrier = lane < LI.first + 2;
        else carrier = lane == LI.first;
        if (carrier) s_z[erow_l * BL + s] = r;
      }
      const double kr = wsum<SMALL>(K * r);
      const double coef = rl(ef_l, s) - kr;
      r = vecTt<GLOB>(B, LI, r, lane, mv);
      retreat(B, LI, mv, lane);
      // + Z coef (layout of t)
      if (LI.template zsel<GLOB>(lane)) r += coef;
      if (!mylane) r = 0.0;
    }
    wave_lds_sync();
    for (int e = 0; e < NE; ++e)
      if (in_l) gd[(size_t)e * T + tt] = s_z[e * BL + lane];
    wave_lds_sync();
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  }
  SSTAMP(6);
  // ---- 5. forward: the mean correction E(alpha | y) - E(alpha | y+), the state
  // draw, the state models' and the regression's sufficient statistics -- by BOTH waves:
  // wave 0 runs the recursion of the correction and turns a block of alpha+ (LDS) into the
  // block of state draws; wave 1 (its variance pass long over) follows one block behind with
  // everything that only READS the draws: the state models' sufficient statistics, the
  // residuals, the copy in logical order that goes out.  The two block buffers take turns.
  if (wave == 0) {
    double mc = P0l * r;          // a0 + P0 r0 - (a0 + P0 r0+)
    seek(B, LI, 0, -1);
    for (int tb = 0; tb < T; tb += BL) {
      const int tt = tb + lane, bi = tb / BL;
      const int nstep = (T - tb < BL) ? T - tb : BL;
      const bool in_l = lane < nstep;
      double *buf = (bi & 1) ? s_blk1 : s_blk0;
      while (__hip_atomic_load(&s_cdone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < bi - 1)
        __builtin_amdgcn_s_sleep(4);
      blk_load(buf, gst + (size_t)tb * m, nstep * m, lane);
      // r_{t-1} at the error rows, for the steps into tb .. tb + nstep - 1
      for (int e = 0; e < NE; ++e) s_z[e * BL + lane] = (in_l && tt > 0) ? gd[(size_t)e * T + tt - 1] : 0.0;
      wave_lds_sync();
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double ap = mylane ? buf[s * m + lane] : 0.0;
        if (tb + s > 0) {
          const unsigned mv = B.moving();
          mc = vecT<SMALL, GLOB>(B, LI, mc, lane, mv);
          advance(B, LI, mv, lane);
          // + RQR_{t-1} r_{t-1}
          bool carrier = false;
          if (mylane) {
            if (LI.kind == SSG_SEASONAL) carrier = LI.moves(mv) && lane == LI.first + LI.cur;
            else if (LI.kind == SSG_LOCAL_LINEAR_TREND || (GLOB && LI.kind == SSG_TRIG)) carrier = true;
            else if (GLOB && LI.kind == SSG_SEMILOCAL) carrier = lane < LI.first + 2;
            else carrier = lane == LI.first;
          }
          if (carrier) mc += sig_l * s_z[erow_l * BL + s];
        } else {
          advance(B, LI, 0u, lane);
        }
        if (mylane) buf[s * m + lane] = ap + mc;
      }
      wave_lds_sync();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&s_cprog, bi + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
  double suf_l = 0.0;           // local level / seasonal: sum of squared state errors (at the lane that carried them)
  double mv_ybar = 0.0, mv_sumsq = 0.0, mv_n = 0.0;   // MvnSuf of a trend block's errors (its two lanes)
  double yty = 0.0, nobs = 0.0;
  // ArModel's NeRegSuf of now[first] on then[first ..]: lane first + i keeps xty_i and row i
  // of xtx, the row in LDS at s_axx[(block's slot * AR_MAX + i) * (AR_MAX + 1) + q]
  double axy = 0.0, ayy = 0.0;
  // a semilocal trend's Ar1Suf of the slope draws (its lane first + 1): Ar1Suf::update_raw
  double a1_sumsq = 0.0, a1_sum = 0.0, a1_cross = 0.0, a1_first = 0.0, a1_last = 0.0;
  for (int e2 = lane; e2 < M.nar * AR_MAX * (AR_MAX + 1); e2 += WAVE) s_axx[e2] = 0.0;
  wave_lds_sync();
  seek(B, LI, 0, -1);
  {
    for (int tb = 0; tb < T; tb += BL) {
      const int tt = tb + lane, bi = tb / BL;
      const int nstep = (T - tb < BL) ? T - tb : BL;
      const bool in_l = lane < nstep;
      double *buf = (bi & 1) ? s_blk1 : s_blk0;
      const double y_l = in_l ? P.y[tt] : 0.0;
      const int ob_l = (in_l && P.observed[tt]) ? 1 : 0;
      // (wave 0 only leaves early when THIS wave's variance pass failed, and then this wave has left too)
      while (__hip_atomic_load(&s_cprog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= bi)
        __builtin_amdgcn_s_sleep(8);
      double res_l = 0.0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double st = mylane ? buf[s * m + lane] : 0.0;
        unsigned mv = 0;
        double tot_then = 0.0;   // seasonal: sum of the block at t - 1 (kept by the lanes of the block)
        if (tb + s > 0) {
          mv = B.moving();
          // (the seasonal models' observe_state needs the sum of `then` over the block)
          unsigned sm = mv & B.seasmask;
          while (sm) {
            const int b = __ffs((int)sm) - 1;
            sm &= sm - 1;
            const double tb_ = wsum<SMALL>(LI.blk == b ? prev : 0.0);
            if (LI.blk == b) tot_then = tb_;
          }
          advance(B, LI, mv, lane);
        } else {
          advance(B, LI, 0u, lane);
        }
        const double then1 = from_above(prev);
        const double thenb = (GLOB && B.trigmask) ? from_below(prev) : 0.0;
        if (GLOB && LI.kind == SSG_SEMILOCAL && lane == LI.first + 1) {
          // observe_initial_state / observe_state: the current slope into the Ar1Suf, every t
          // (SemilocalLinearTrend.cpp:168-180; NonzeroMeanAr1Model.cpp:39-49)
          if (tb + s == 0) a1_first = st; else a1_cross += st * a1_last;
          a1_sum += st;
          a1_sumsq += st * st;
          a1_last = st;
        }
        if (tb + s > 0) {
          if (LI.kind == SSG_LOCAL_LEVEL) {
            const double diff = st - prev;
            suf_l += diff * diff;
          } else if (GLOB && LI.kind == SSG_SEMILOCAL) {
            if (lane == LI.first) {
              const double change_in_level = st - prev - then1;
              suf_l += change_in_level * change_in_level;
            }
          } else if (GLOB && LI.kind == SSG_TRIG) {
            // now - rotation * then, every component (TrigStateModel::observe_state, TrigStateModel.cpp:182-193)
            const double rot = LI.todd(lane) ? -LI.ts * thenb + LI.tc * prev : LI.tc * prev + LI.ts * then1;
            const double e = st - rot;
            suf_l += e * e;
          } else if (LI.kind == SSG_LOCAL_LINEAR_TREND) {
            // err = now - T then; MvnSuf::update_raw (MvnBase.cpp:71-86), diagonal only
            const double err = st - ((lane == LI.first) ? prev + then1 : prev);
            mv_n += 1.0;
            const double wv = (err - mv_ybar) / mv_n;
            mv_ybar += wv;
            mv_sumsq += wv * wv * (mv_n - 1);
            const double w2 = err - mv_ybar;
            mv_sumsq += w2 * w2;
          } else if (LI.kind == SSG_SEASONAL) {
            // delta = now[0] + sum(then) over the block, on the steps into a new season
            if (LI.moves(mv) && lane == LI.first + LI.cur) {
              const double dl = st - (-1.0 * tot_then);
              suf_l += dl * dl;
            }
          }
          // autoregression: add_mixture_data(now[0], then, 1.0): xtx += then then', xty += now[0] then, yty += now[0]^2
          unsigned am = B.armask;
          while (am) {
            const int b = __ffs((int)am) - 1;
            am &= am - 1;
            const unsigned d = B.udesc(b);
            const int f = Blocks::first_of(d), n = Blocks::dim_of(d);
            const double yy = rl(st, f);
            double *rowx = s_axx + ((size_t)Blocks::arx_of(d) * AR_MAX + (LI.blk == b ? lane - f : 0)) * (AR_MAX + 1);
#pragma nounroll
            for (int q = 0; q < n; ++q) {
              const double pq = rl(prev, f + q);
              if (LI.blk == b) rowx[q] += prev * pq * 1.0;
            }
            if (LI.blk == b) {
              axy += (yy * 1.0) * prev;
              ayy += yy * yy * 1.0;
            }
          }
        }
        prev = st;
        // the state draw goes out in logical order (in place: every lane has read its entry)
        if (mylane && LI.kind == SSG_SEASONAL) {
          const int q = lane - LI.first, c = LI.cur;
          buf[s * m + LI.first + (q >= c ? q - c : q - c + LI.dim)] = st;
        }
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const double resid = obs ? rl(y_l, s) - zdot<SMALL, GLOB>(LI, st, lane) : 0.0;
        if (lane == s) res_l = resid;
        if (obs) { yty += resid * resid; nobs += 1.0; }
      }
      blk_store(gst + (size_t)tb * m, buf, nstep * m, lane);
      if (in_l) sres[tt] = res_l;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&s_cdone, bi + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  // publish the sufficient statistics
  for (int b = 0; b < nb; ++b) {
    const unsigned d = B.udesc(b);
    const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d);
    const size_t at = (size_t)chain * SSG_MAX_VAR + Blocks::var0_of(d);
    if (kd == SSG_LOCAL_LEVEL) {
      if (lane == f) {
        M.var_n[at] = (double)(T - 1);
        M.var_ss[at] = suf_l;
      }
    } else if (kd == SSG_LOCAL_LINEAR_TREND) {
      // center_sumsq(mu = 0)(i, i) = sumsq_ii + n ybar_i^2
      const double ssv = mv_sumsq + mv_ybar * mv_ybar * mv_n;
      if (lane == f || lane == f + 1) {
        M.var_n[at + (lane - f)] = mv_n;
        M.var_ss[at + (lane - f)] = ssv;
      }
    } else if (kd == SSG_SEMILOCAL) {
      if (lane == f) {
        M.var_n[at] = (double)(T - 1);
        M.var_ss[at] = suf_l;
      }
      if (lane == f + 1) {
        double *suf = M.ar_suf + ((size_t)chain * SSG_MAX_AR + Blocks::arx_of(d)) * AR_SUF_STRIDE;
        suf[0] = a1_sumsq; suf[1] = a1_sum; suf[2] = a1_cross; suf[3] = (double)T; suf[4] = a1_first; suf[5] = a1_last;
        M.var_n[at + 1] = (double)T;
        M.var_ss[at + 1] = 0.0;
      }
    } else if (kd == SSG_TRIG) {
      // one GaussianSuf for all the block's components
      const double tot = wsum<SMALL>(LI.blk == b ? suf_l : 0.0);
      if (lane == f) {
        M.var_n[at] = (double)n * (double)(T - 1);
        M.var_ss[at] = tot;
      }
    } else if (kd == SSG_SEASONAL) {
      // (the lane that accumulated moved with the cursor: sum over the block)
      const double tot = wsum<SMALL>(LI.blk == b ? suf_l : 0.0);
      const unsigned dpw = (unsigned)__builtin_amdgcn_readlane((int)B.dp, b);
      if (lane == f) {
        M.var_n[at] = (double)seasons_started(T - 1, (int)(dpw & 0xffffu), (int)(dpw >> 16));
        M.var_ss[at] = tot;
      }
    } else {
      double *suf = M.ar_suf + ((size_t)chain * SSG_MAX_AR + Blocks::arx_of(d)) * AR_SUF_STRIDE;
      if (LI.blk == b) {
        const int i = lane - f;
        const double *rowx = s_axx + ((size_t)Blocks::arx_of(d) * AR_MAX + i) * (AR_MAX + 1);
        for (int q = 0; q < n; ++q) suf[i * AR_MAX + q] = rowx[q];
        suf[AR_SUF_XTY + i] = axy;
      }
      if (lane == f) {
        suf[AR_SUF_YTY] = ayy;
        suf[AR_SUF_N] = (double)(T - 1);
      }
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
// advances by T state + state errors (model by model), the observation is
// rnorm(Z'state, sigma_obs) + x'beta; normals in the reference's order on the
// chain's forecast stream (id 5).  One wavefront per chain, lane = state component
// (logical order; the horizon is short, a seasonal block simply shifts).  As the
// reference (advance_to_timestamp, StateSpaceModelBase.cpp:455-459) forecast step i
// uses the transition matrix and state errors of time T - 2 + i.
__global__ __launch_bounds__(64) void ssg_forecast_kernel(SsParams P, int horizon, const double *newX,
                                                          uint64_t *pos_forecast, double *out) {
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) return;
  const SsmParams &M = P.ssm;
  const SsgSpec &Q = *M.spec;
  const int T = P.T, p = P.p, m = M.m, nb = M.nblocks;
  const double *beta = P.beta + (size_t)chain * p;
  const double sd_obs = sqrt(P.sigsq[chain]);
  const double *gst = M.work + (size_t)chain * M.work_stride + (size_t)m * T;
  double st = (lane < m) ? gst[(size_t)(T - 1) * m + lane] : 0.0;
  SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 5u}, pos_forecast[chain]};
  for (int i = 0; i < horizon; ++i) {
    const int tm = T - 2 + i;   // the transition's index
    double nx = st;
    for (int b = 0; b < nb; ++b) {
      const SsgBlock &K = Q.blk[b];
      const int f = K.first, n = K.dim;
      const bool mine = lane >= f && lane < f + n;
      const double *sg = M.var_sigsq + (size_t)chain * SSG_MAX_VAR + K.var0;
      if (K.kind == SSG_LOCAL_LEVEL) {
        const double e0 = d_rnorm(rng, 0.0, sqrt(sg[0]));
        if (lane == f) nx = st + e0;
      } else if (K.kind == SSG_LOCAL_LINEAR_TREND) {
        const double z0 = d_rnorm(rng, 0.0, 1.0), z1 = d_rnorm(rng, 0.0, 1.0);
        const double x1 = rl(st, f + 1);
        if (lane == f) nx = (st + x1) + (sqrt(sg[0]) * z0 + 0.0);
        if (lane == f + 1) nx = st + (sqrt(sg[1]) * z1 + 0.0);
      } else if (K.kind == SSG_SEMILOCAL) {
        const double *ph = M.ar_phi + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_MAX;
        const double e0 = d_rnorm(rng, 0.0, sqrt(sg[0])), e1 = d_rnorm(rng, 0.0, sqrt(sg[1]));
        const double above = from_above(st);
        if (lane == f) nx = (st + above) + e0;
        else if (lane == f + 1) nx = (ph[0] * st + (1 - ph[0]) * above) + e1;
      } else if (K.kind == SSG_TRIG) {
        // rnorm_mt(rng, 0, sigma) per component, in order (TrigStateModel.cpp:218-223), on the rotated state
        const double sd = sqrt(sg[0]);
        const double above = from_above(st), below = from_below(st);
        const double c = mine ? Q.trig_c[lane] : 0.0, sn = mine ? Q.trig_s[lane] : 0.0;
        double e4 = 0.0;
        for (int q = 0; q < n; ++q) {
          const double eq = d_rnorm(rng, 0.0, sd);
          if (lane == f + q) e4 = eq;
        }
        if (mine) nx = (((lane - f) & 1) ? -sn * below + c * st : c * st + sn * above) + e4;
      } else if (K.kind == SSG_SEASONAL) {
        if ((tm + 1) % K.duration == K.phase) {
          const double e2 = d_rnorm(rng, 0.0, sqrt(sg[0]));
          // (first = 0 - s_0 - s_1 - ..., SeasonalStateSpaceMatrix::multiply)
          double firstv = 0.0;
          for (int q = 0; q < n; ++q) firstv -= rl(st, f + q);
          const double below = from_below(st);
          if (lane == f) nx = firstv + e2; else if (mine) nx = below;
        }
      } else {
        const double *ph = M.ar_phi + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_MAX;
        const double e3 = d_rnorm(rng, 0.0, 1.0) * sqrt(sg[0]);
        // (first = sum of phi_i s_i from the last lag down, AutoRegressionTransitionMatrix::multiply_inplace)
        double firstv = 0.0;
        for (int q = n - 1; q >= 0; --q) firstv += ph[q] * rl(st, f + q);
        const double below = from_below(st);
        if (lane == f) nx = firstv + e3; else if (mine) nx = below;
      }
    }
    st = (lane < m) ? nx : 0.0;
    // Z'state: the blocks' first components (a trig block: every pair's first), in state order
    double zs = 0.0;
    for (int b = 0; b < nb; ++b) {
      const SsgBlock &K = Q.blk[b];
      for (int i = 0; i < (K.kind == SSG_TRIG ? K.dim : 1); i += 2) {
        const double zv = rl(st, K.first + i);
        zs = (b == 0 && i == 0) ? zv : zs + zv;
      }
    }
    const double obs = d_rnorm(rng, zs, sd_obs);
    double part = 0.0;
    for (int j = lane; j < p; j += WAVE) part += newX[(size_t)j * horizon + i] * beta[j];
    const double pred = wsum<false>(part);
    if (lane == 0) out[(size_t)chain * horizon + i] = obs + pred;
  }
  if (lane == 0) pos_forecast[chain] = rng.pos;
}

hipError_t launch_ssm_forecast(hipStream_t stream, const SsParams &P, int horizon, const double *newX,
                               uint64_t *pos_forecast, double *out) {
  hipLaunchKernelGGL(ssg_forecast_kernel, dim3(P.chain_count), dim3(WAVE), 0, stream, P, horizon, newX,
                     pos_forecast, out);
  return hipGetLastError();
}

hipError_t launch_xte_tiled(hipStream_t stream, const double *U, int64_t ldu, int R, const double *B, int64_t n,
                            int p, double *out, double *planes);

size_t ssm_dynamic_lds(const SsmParams &M) {
  size_t need = (size_t)ssg_pass_lds_doubles(M.m, M.ld, M.bl, M.nerr, M.nar) * sizeof(double);
  if (need < sizeof(NormalsLds)) need = sizeof(NormalsLds);
  if (need < sizeof(ArLds)) need = sizeof(ArLds);
  return (need + 15) & ~(size_t)15;
}

hipError_t launch_ssm_template(hipStream_t stream, const SsParams &P, int draw_variances);

hipError_t launch_ssm_simsmooth(hipStream_t stream, const SsParams &P, int draw_variances) {
  const dim3 grid(P.chain_count), block(2 * WAVE);
  const size_t lds = ssm_dynamic_lds(P.ssm);
  hipError_t err;
  {
    KtScope kt(stream, KT_SSM);
    if (P.ssm.tpl_trend > 0) {
      err = launch_ssm_template(stream, P, draw_variances);
      if (err != hipSuccess) return err;
    } else {
      // (more than 64 KB of dynamic LDS has to be asked for -- per device, so every time)
      auto go = [&](auto kernel) -> hipError_t {
        if (lds > 65536) {
          const hipError_t e2 = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (e2 != hipSuccess) return e2;
        }
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, P, draw_variances);
        return hipSuccess;
      };
      const bool glob = P.ssm.glob != 0;   // (a trig or semilocal block in the list)
      switch (P.ssm.ld) {   // ssg_leading_dimension(m)
        case 17: err = glob ? go(ssg_simsmooth_kernel<true, 17, true>) : go(ssg_simsmooth_kernel<true, 17, false>); break;
        case 33: err = glob ? go(ssg_simsmooth_kernel<false, 33, true>) : go(ssg_simsmooth_kernel<false, 33, false>); break;
        case 61: err = glob ? go(ssg_simsmooth_kernel<false, 61, true>) : go(ssg_simsmooth_kernel<false, 61, false>); break;
        case 65: err = glob ? go(ssg_simsmooth_kernel<false, 65, true>) : go(ssg_simsmooth_kernel<false, 65, false>); break;
        default: return hipErrorInvalidValue;
      }
      if (err != hipSuccess) return err;
    }
    err = hipGetLastError();
  }
  if (err != hipSuccess) return err;
  // xty[chain, j] = x_j' e_chain (the residual series are array 1 of every chain's scratch block)
  return launch_xte_tiled(stream, P.scratch + (size_t)P.chain_first * P.scratch_stride + P.T, P.scratch_stride,
                          P.chain_count, P.X, (int64_t)P.T, P.p, P.xty + (size_t)P.chain_first * P.p,
                          P.xte_planes);
}

}  // namespace boom_amd
