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
// State dimension 1 (Z = 1, T = 1, RQR = sigma^2_level): every per-time-step
// quantity is a scalar, and given the gains K_t every recursion of the
// reference is an affine map x -> A_t x + B_t applied in time order.  The
// kernel therefore works on the time axis 64 steps at a time, lane = step:
//   1. y*_t = y_t - x_t'beta            lane-parallel, coalesced X columns
//   2. the sweep's standard normals     one substream position per draw (state
//                                       error, then observation, for every t:
//                                       stream_normals.h)
//   3. P_t, F_t, K_t                    a Riccati recursion, i.e. a Moebius map of
//                                       P_t / H: a scan of 2 x 2 matrices
//   4. forward: simulated states (prefix sum), and ONE filter on
//      w_t = y*_t - y+_t: the data filter and the simulation filter share K_t,
//      so their difference delta_t = a_t - a+_t obeys
//      delta_{t+1} = (1 - K_t) delta_t + K_t w_t (a wave scan of affine maps)
//   5. backward: d_{t-1} = e_t / F_t + (1 - K_t) d_t for d = r - r+ (same scan,
//      lanes reversed)
//   6. forward: mean correction (prefix sum of q d_{t-1}), state_t, level suf
//   7. residuals, X'e, e'e             lane-parallel
// The reference runs the two filters / smoothers separately and subtracts at
// the end; by linearity the difference recursion gives the same state draw up
// to rounding (parity tolerance in tests/test_state_space_gpu.py).
#include <hip/hip_runtime.h>

#include "ktimer.h"

#include "kalman_lm_device.h"

namespace boom_amd {

namespace {

enum : int { NR = 16 };  // registers per lane of a time panel

// In the three passes a lane owns CONSECUTIVE time steps, so a plain load of "my
// j-th step" touches 64 different cache lines per instruction -- the passes were bound
// by exactly that (24 such loads and as many stores per chunk and wave).  Instead a
// wave moves its stretch between HBM and registers through LDS: coalesced rows of 64
// (8 lines per instruction), transposed in a padded staging buffer (element e at
// e + e / 32: a half-wave's 32 row elements sit in 32 different banks, and so do the
// j-th elements of 32 lanes' stretches of 8 -- 8 l + l / 4 + j mod 32 is one-to-one in l).  reverse: the lane's j-th value is element 64 CNT - 1 - (CNT lane + j) (the
// backward pass walks time downwards).
__device__ __forceinline__ int stage_at(int e) { return e + (e >> 5); }
enum : int { STAGE_DOUBLES = 2 * 8 * WAVE + 2 * 8 * WAVE / 32 };   // room for CNT = 16

// raw[i] = g[first + 64 i + lane] where that index is in [0, n), else fill: the
// coalesced half of a block load (all of a chunk's are issued before any transpose, so
// that their latencies overlap)
template <int CNT>
__device__ __forceinline__ void wave_block_fetch(const double *__restrict__ g, int64_t first, int64_t n,
                                                 int lane, double fill, double (&raw)[CNT]) {
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int64_t idx = first + i * WAVE + lane;
    raw[i] = (idx >= 0 && idx < n) ? g[idx] : fill;
  }
}
// ... and the transpose: out[j] = element CNT lane + j of the block (reverse: counted
// from its end)
template <int CNT>
__device__ __forceinline__ void wave_block_turn(double *stage, int lane, bool reverse,
                                                const double (&raw)[CNT], double (&out)[CNT]) {
#pragma unroll
  for (int i = 0; i < CNT; ++i) stage[stage_at(i * WAVE + lane)] = raw[i];
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int e = reverse ? (WAVE * CNT - 1 - (CNT * lane + j)) : (CNT * lane + j);
    out[j] = stage[stage_at(e)];
  }
  wave_lds_sync();
}
// g[first + CNT lane + j] = v[j] where that index is in [0, n)
template <int CNT>
__device__ __forceinline__ void wave_block_store(double *stage, double *__restrict__ g, int64_t first,
                                                 int64_t n, int lane, bool reverse, const double (&v)[CNT]) {
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int e = reverse ? (WAVE * CNT - 1 - (CNT * lane + j)) : (CNT * lane + j);
    stage[stage_at(e)] = v[j];
  }
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int e = i * WAVE + lane;
    const int64_t idx = first + e;
    if (idx >= 0 && idx < n) g[idx] = stage[stage_at(e)];
  }
  wave_lds_sync();
}

}  // namespace

// grid = chains, block = 128: both waves share every phase (y* panels, the sweep's
// normals, the time-blocked scans of the three passes); wave 0 publishes the results.
__global__ __launch_bounds__(128) void kalman_simsmooth_kernel(SsParams P,
                                                               int draw_level) {
  union KalmanLds {
    NormalsLds norm;                        // the normals generator's lists, then ...
    double stage[2][STAGE_DOUBLES];         // ... the passes' staging buffers, one per wave
  };
  __shared__ KalmanLds s_u;
  NormalsLds &s_norm = s_u.norm;
  __shared__ double s_x[2][8];             // the two waves' scan totals, swapped at the seam between their stretches
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= P.chain_count) return;
  int32_t *prep_n_slot = P.prep_n + (size_t)P.zbuf * P.chains + chain;
  const uint64_t *prep_pos_slot = P.prep_pos_state + (size_t)P.zbuf * P.chains + chain;
  const int prep_n = *prep_n_slot, status_in = P.status[chain];
  const int ran = P.only_ran ? P.only_ran[chain] : 1;
  const bool prepared = P.prepared != 0 && prep_n > 0;
  __syncthreads();   // (every thread has read the chain's words before thread 0 rewrites any: kalman_lm_device.h)
  if (P.prepared != 0 && prep_n < 0 && status_in == CHAIN_OK) {   // the prepare step failed
    if (threadIdx.x == 0) { P.status[chain] = -prep_n; *prep_n_slot = 0; }
    return;
  }
  if (status_in != CHAIN_OK || ran == 0) {
    // a chain that sits this round out: what kalman_prepare_kernel did ahead for it is undone
    if (prepared && threadIdx.x == 0) {
      P.pos_state[chain] = *prep_pos_slot;
      P.pos_level[chain] = P.prep_pos_level[(size_t)P.zbuf * P.chains + chain];
      P.level_sigsq[chain] = P.prep_level_sigsq[(size_t)P.zbuf * P.chains + chain];
      *prep_n_slot = 0;
    }
    return;
  }
  const int T = P.T, p = P.p;
  int status = CHAIN_OK;
#ifdef BA_KSTAMPS
  long long kph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, klast = (long long)__builtin_readcyclecounter();
#endif

  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  double level_sigsq = P.level_sigsq[chain];

  // ---- ZeroMeanGaussianConjSampler::draw: sigma^2_level | state
  if (draw_level && !prepared) {
    SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 1u}, P.pos_level[chain]};
    int bad = 0;
    const double DF = P.level_n[chain] + P.level_prior_df;
    const double SS = P.level_sumsq[chain] + P.level_prior_ss;
    level_sigsq = d_draw_variance(rng, DF, SS, P.level_sigma_max, &bad);
    if (bad) status = CHAIN_RNG_BRANCH;
    __syncthreads();   // (every thread has read the position it draws from: see ssm_template_kernel.hip)
    if (lane == 0 && wave == 0) {  // (wave 1 repeats the draw: it needs sigma_level != 0)
      P.pos_level[chain] = rng.pos;
      P.level_sigsq[chain] = level_sigsq;
    }
  }
  if (status != CHAIN_OK) {
    if (lane == 0 && wave == 0) P.status[chain] = status;
    return;
  }
  if (P.level_used && lane == 0 && wave == 0) P.level_used[chain] = level_sigsq;

  // ---- impute_state ------------------------------------------------------
  const double sigsq_obs = P.sigsq[chain];
  const double *beta = P.beta + (size_t)chain * p;
  double *w0 = P.scratch + (size_t)chain * P.scratch_stride;  // y* -> e/F -> d -> residual
  double *sF = w0 + T;                                        // residuals y - state (input of the X'e GEMM)
  double *sK = sF + T;                                        // K_t
  double *sal = sK + T;                                       // simulated state alpha+_t
  double *sst = sal + T;                                      // the state draw (SS_STATE_ARRAY)
  double *szz = sst + T + (size_t)P.zbuf * 2 * T;             // the sweep's normals, stream order (2 T; two buffers)

  const double q = level_sigsq;
  const double level_sigma = sqrt(level_sigsq);
  const double H = sigsq_obs;  // one observation per time point (n_t = 1)
  const double sqrtH = sqrt(H);
  const double sd0 = sqrt(P.P0);

  KSTAMP(0);
  // ---- 1. adjusted observations y*_t = y_t - x_t'beta
  // (StateSpaceRegressionModel.cpp:65-77, 179-181; GlmCoefs::predict is a dense
  // dot with Beta(), zeros outside gamma).  Not a pass of its own any more: the forward
  // pass computes its chunk's y* where it used to fetch it (ystar_block below) -- one
  // array less written and read back per chain and sweep.
  KSTAMP(1);
  // ---- 2. the standard normals of simulate_forward, in stream order:
  // t = 0: initial state (if P0 > 0), observation (if sigma_obs > 0);
  // t >= 1: state error (if sigma_level > 0), observation.  rnorm_mt draws
  // nothing when its sigma is 0 (Bmath/rnorm.cpp:63-64).
  const int dI = (sd0 != 0.0), dL = (level_sigma != 0.0), dH = (sqrtH != 0.0);
  const int nfirst = dI + dH, nper = dL + dH;
  const int N = nfirst + (T - 1) * nper;
  // Normal i of the sweep reads its uniforms from position bpos0 + 256 i of the
  // chain's state stream (stream_normals.h); szz holds them in draw order.
  if (!prepared || prep_n != N) {
    // (prepared with another count: the observation variance turned out to be exactly
    // zero, which the prepare step cannot know -- the normals again, from where it
    // started; the next prepare step goes out behind this kernel, so the stream position
    // written here is the one it reads)
    const uint64_t bpos0 = prepared ? *prep_pos_slot : P.pos_state[chain];
    status = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, bpos0, N, szz,
                            &P.pos_state[chain], ss_slot_serve(P));
    if (status != CHAIN_OK) {   // (s_hand[3]: the same in both waves)
      if (threadIdx.x == 0) P.status[chain] = status;
      return;
    }
  }
  __syncthreads();
  if (prepared && threadIdx.x == 0) *prep_n_slot = 0;   // consumed
  double *stage = s_u.stage[wave];   // (the lists are done with)
  KSTAMP(2);
  // ---- 3 + 4. forward pass.
  // Variances (ScalarMarginalDistribution::update, the part that does not look
  // at the data): P_{t+1} = P_t - P_t^2 / (P_t + H) + q at an observed step,
  // P_t + q at a missing one.  In units of H (u = P / H, r = q / H) that is the
  // Moebius map u -> ((1 + r) u + r) / (u + 1), resp. u + r, so its composition
  // over any stretch of time is a product of 2 x 2 matrices (divided by 1 + r:
  // determinant 1, entries <= 1, no over- or underflow) applied to the incoming
  // u.  Then F_t = H (u_t + 1), K_t = u_t / (u_t + 1); simulate_initial_state /
  // simulate_next_state (alpha+), simulate_adjusted_observation (y+), and the
  // filter on w = y* - y+.
  //
  // Every pass below gives a lane BS CONSECUTIVE time steps and a wave 64 BS of
  // them; a chunk is the two waves' stretches side by side (wave 0 first in scan
  // order).  A lane composes its own steps' maps serially, ONE wave scan combines
  // the lanes' composites, the two waves swap their totals through LDS (one
  // barrier), and each lane walks its steps again from the value this hands it.
  // That is ~1/BS of the scan work of a lane-per-step pass, on both waves.
  KSTAMP(3);
  constexpr int BS = 8, WS = WAVE * BS, CS = 2 * WS;   // steps per lane, per wave, per chunk
  {
    const bool moebius = H > 0.0;  // (H == 0: the plain recursion, lane after lane)
    const double r = moebius ? q / H : 0.0, s1 = 1.0 / (1.0 + r);
    double u_in = moebius ? P.P0 / H : P.P0;   // variance carried in units of H (of 1 when H == 0)
    double alpha_in = 0.0, delta_in = 0.0;
    for (int t0 = 0; t0 < T; t0 += CS) {
      const int tl = t0 + WS * wave + BS * lane;
      bool in[BS], obs[BS];
      double zL[BS], zH[BS], ys[BS];
      const int tw = t0 + WS * wave;   // the wave's first step
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int t = tl + j;
        in[j] = t < T;
        obs[j] = (int)in[j] & (int)(P.observed[in[j] ? t : 0] != 0);   // (& not &&: the load unconditional, all eight together)
      }
      // the normals of steps t >= 1 sit nper to a step from nfirst on: state error (if
      // any), then observation error (if any)
      {
        const int64_t gfirst = (int64_t)nfirst + (int64_t)(tw - 1) * nper;
        // y* of the wave's stretch in the fetch layout (element i = step tw + 64 i + lane, so
        // that a variable's column is BS coalesced loads): included variables in batches of
        // 64, lane m of a batch holding (j_m, beta_m)
        double yraw[BS];
        {
          double pred[BS];
#pragma unroll
          for (int i = 0; i < BS; ++i) pred[i] = 0.0;
          for (int base = 0; base < p; base += WAVE) {
            const int jv = base + lane;
            const double bj = (jv < p) ? beta[jv] : 0.0;
            unsigned long long mk = __ballot(bj != 0.0);
            while (mk) {
              const int l = __ffsll((long long)mk) - 1;
              mk &= mk - 1;
              const double b = bcast_u(bj, l);
              // (branch-free: steps past T read the last row and are never used)
              const double *col = P.X + (size_t)(base + l) * T;
              double xv[BS];
#pragma unroll
              for (int i = 0; i < BS; ++i) {
                const int t = tw + i * WAVE + lane;
                xv[i] = col[t < T ? t : T - 1];
              }
#pragma unroll
              for (int i = 0; i < BS; ++i) pred[i] += xv[i] * b;
            }
          }
          // (unconditional, all eight together, and the compiler told so: as "in range ? y[t] -
          // ... : 0" every load had a branch and a wait of its own; tools/isa_serial_loads.py)
          double yt[BS];
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const int t = tw + i * WAVE + lane;
            yt[i] = P.y[t < T ? t : 0];
          }
          static_assert(BS == 8, "the list below");
          asm volatile("" : "+v"(yt[0]), "+v"(yt[1]), "+v"(yt[2]), "+v"(yt[3]), "+v"(yt[4]), "+v"(yt[5]), "+v"(yt[6]), "+v"(yt[7]));
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const int t = tw + i * WAVE + lane;
            yraw[i] = (t < T) ? yt[i] - pred[i] : 0.0;
          }
        }
        if (nper == 2) {
          double zraw[2 * BS], zz[2 * BS];
          wave_block_fetch<2 * BS>(szz, gfirst, N, lane, 0.0, zraw);
          wave_block_turn<BS>(stage, lane, false, yraw, ys);
          wave_block_turn<2 * BS>(stage, lane, false, zraw, zz);
#pragma unroll
          for (int j = 0; j < BS; ++j) { zL[j] = zz[2 * j]; zH[j] = zz[2 * j + 1]; }
        } else if (nper == 1) {
          double zraw[BS], z1[BS];
          wave_block_fetch<BS>(szz, gfirst, N, lane, 0.0, zraw);
          wave_block_turn<BS>(stage, lane, false, yraw, ys);
          wave_block_turn<BS>(stage, lane, false, zraw, z1);
#pragma unroll
          for (int j = 0; j < BS; ++j) { zL[j] = dL ? z1[j] : 0.0; zH[j] = dL ? 0.0 : z1[j]; }
        } else {
          wave_block_turn<BS>(stage, lane, false, yraw, ys);
#pragma unroll
          for (int j = 0; j < BS; ++j) { zL[j] = 0.0; zH[j] = 0.0; }
        }
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          if (!in[j]) { zL[j] = 0.0; zH[j] = 0.0; }
        }
        if (tl == 0) {   // step 0: initial state (if any), then observation error (if any)
          zL[0] = dI ? szz[0] : 0.0;
          zH[0] = dH ? szz[dI] : 0.0;
        }
      }
      // ---- what does not depend on the carries: the lane's composite variance
      // map and its partial sums of the simulated state's increments
      Mob M;
      M.a = 1.0; M.b = 0.0; M.c = 0.0; M.d = 1.0;
      double al[BS], asum = 0.0;
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        Mob mt;
        mt.a = 1.0;
        mt.b = obs[j] ? r * s1 : (in[j] ? r : 0.0);
        mt.c = obs[j] ? s1 : 0.0;
        mt.d = obs[j] ? s1 : 1.0;
        M = mob_after(mt, M);
        // alpha+_0 = rnorm(a0, sqrt(P0)); alpha+_t = alpha+_{t-1} + rnorm(0, sigma_level)
        const int t = tl + j;
        asum += !in[j] ? 0.0 : ((t == 0) ? P.a0 + sd0 * zL[j] : level_sigma * zL[j]);
        al[j] = asum;
      }
      Mob G;
      G.a = 1.0; G.b = 0.0; G.c = 0.0; G.d = 1.0;
      if (moebius) G = wave_scan(M);
      const double aincl = wave_prefix_sum(asum);
      if (lane == WAVE - 1) {
        s_x[wave][0] = G.a; s_x[wave][1] = G.b; s_x[wave][2] = G.c; s_x[wave][3] = G.d;
        s_x[wave][4] = aincl;
      }
      __syncthreads();
      Mob G0, G1;   // the two waves' totals
      G0.a = s_x[0][0]; G0.b = s_x[0][1]; G0.c = s_x[0][2]; G0.d = s_x[0][3];
      G1.a = s_x[1][0]; G1.b = s_x[1][1]; G1.c = s_x[1][2]; G1.d = s_x[1][3];
      const double a0tot = s_x[0][4], a1tot = s_x[1][4];
      // ---- variances: K_t = u_t / (u_t + 1), so u_{t+1} = K_t + r
      double K[BS], Fv[BS];
      if (moebius) {
        Mob E;  // the lanes before this one (in this wave, and all of wave 0 for wave 1)
        E.a = lane_before(G.a, 1.0);
        E.b = lane_before(G.b, 0.0);
        E.c = lane_before(G.c, 0.0);
        E.d = lane_before(G.d, 1.0);
        if (wave == 1) E = mob_after(E, G0);
        double u = (E.a * u_in + E.b) / (E.c * u_in + E.d);
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const double kk = u / (u + 1.0);
          Fv[j] = H * (u + 1.0);
          K[j] = obs[j] ? kk : 0.0;   // 0 at a missing observation: delta stays
          u = obs[j] ? kk + r : (in[j] ? u + r : u);
        }
        const Mob GT = mob_after(G1, G0);
        u_in = (GT.a * u_in + GT.b) / (GT.c * u_in + GT.d);
      } else {
        // lane after lane, wave 0 then wave 1 (both waves run all of it; a wave
        // keeps the values of its own lanes)
        double Pv = u_in;
        for (int hw = 0; hw < 2; ++hw) {
          for (int l = 0; l < WAVE; ++l) {
            // (the stretch of lane l of wave hw: flags come from its owner through LDS)
            const int tb = t0 + WS * hw + BS * l;
            double Pl = Pv;
#pragma unroll
            for (int j = 0; j < BS; ++j) {
              const int t = tb + j;
              const bool inx = t < T;
              const bool obx = (int)inx & (int)(P.observed[inx ? t : 0] != 0);
              const double PZ = Pl, Fi = PZ + H;
              const double Ki = obx ? PZ / Fi : 0.0;
              if (hw == wave && lane == l) { Fv[j] = Fi; K[j] = Ki; }
              if (obx) Pl = Pl + (-1.0) * PZ * Ki;
              if (inx) Pl = Pl + q;
            }
            Pv = Pl;
          }
        }
        u_in = Pv;
      }
      bool badF = false;
#pragma unroll
      for (int j = 0; j < BS; ++j) badF = badF || (in[j] && !(Fv[j] > 0.0));
      const bool anybad = __any(badF) != 0;
      // ---- alpha+ with the carries in
      {
        const double base = alpha_in + (wave == 1 ? a0tot : 0.0) + (aincl - asum);
#pragma unroll
        for (int j = 0; j < BS; ++j) al[j] += base;
        alpha_in += a0tot + a1tot;
      }
      // ---- the filter on w = y* - y+:  delta_{t+1} = (1 - K_t) delta_t + K_t w_t
      double w[BS];
      Aff C;
      C.A = 1.0; C.B = 0.0;
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        w[j] = in[j] ? ys[j] - (al[j] + sqrtH * zH[j]) : 0.0;
        Aff f;
        f.A = 1.0 - K[j];
        f.B = K[j] * w[j];
        C = aff_after(f, C);
      }
      const Aff Gd = wave_scan(C);
      __syncthreads();   // (everyone has read the first exchange)
      if (lane == WAVE - 1) {
        s_x[wave][0] = Gd.A; s_x[wave][1] = Gd.B; s_x[wave][2] = anybad ? 1.0 : 0.0;
      }
      __syncthreads();
      Aff D0, D1;
      D0.A = s_x[0][0]; D0.B = s_x[0][1];
      D1.A = s_x[1][0]; D1.B = s_x[1][1];
      const bool bad_any = (s_x[0][2] != 0.0) || (s_x[1][2] != 0.0);
      __syncthreads();   // (... and the second, before the next chunk overwrites it)
      if (bad_any) { status = CHAIN_FORECAST_VARIANCE; break; }
      Aff Ed;
      Ed.A = lane_before(Gd.A, 1.0);
      Ed.B = lane_before(Gd.B, 0.0);
      if (wave == 1) Ed = aff_after(Ed, D0);
      double delta = Ed.A * delta_in + Ed.B;
      {
        const Aff DT = aff_after(D1, D0);
        delta_in = DT.A * delta_in + DT.B;
      }
      double ef[BS];
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        ef[j] = obs[j] ? (w[j] - delta) / Fv[j] : 0.0;   // (v_t - v+_t) / F_t
        delta = (1.0 - K[j]) * delta + K[j] * w[j];
      }
      wave_block_store<BS>(stage, sal, tw, T, lane, false, al);
      wave_block_store<BS>(stage, sK, tw, T, lane, false, K);
      wave_block_store<BS>(stage, w0, tw, T, lane, false, ef);
    }
  }
  if (status != CHAIN_OK) {
    if (threadIdx.x == 0) P.status[chain] = status;
    return;
  }
  __syncthreads();

  KSTAMP(4);
  // ---- 5. backward: fast_disturbance_smooth for d = r - r+:
  // d_{t-1} = e_t / F_t + (1 - K_t) d_t, d_{T-1} = 0; latest steps first: wave 0
  // has the later half of a chunk, lane l of a wave the BS steps ending at
  // (its wave's end) - BS l - 1
  double d_first;  // d_{-1}
  {
    double d_in = 0.0;
    const int tlast = ((T - 1) / CS) * CS;
    for (int t0 = tlast; t0 >= 0; t0 -= CS) {
      double fa[BS], fb[BS];
      const int tw = t0 + CS - WS * (wave + 1);   // the wave's earliest step
      {
        double kraw[BS], eraw[BS];
        wave_block_fetch<BS>(sK, tw, T, lane, 0.0, kraw);
        wave_block_fetch<BS>(w0, tw, T, lane, 0.0, eraw);
        wave_block_turn<BS>(stage, lane, true, kraw, fa);
        wave_block_turn<BS>(stage, lane, true, eraw, fb);
      }
      Aff C;
      C.A = 1.0; C.B = 0.0;
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        fa[j] = 1.0 - fa[j];   // (1 at steps past T: K reads as 0 there)
        Aff f;
        f.A = fa[j]; f.B = fb[j];
        C = aff_after(f, C);
      }
      const Aff G = wave_scan(C);
      if (lane == WAVE - 1) { s_x[wave][0] = G.A; s_x[wave][1] = G.B; }
      __syncthreads();
      Aff D0, D1;
      D0.A = s_x[0][0]; D0.B = s_x[0][1];
      D1.A = s_x[1][0]; D1.B = s_x[1][1];
      __syncthreads();
      Aff E;
      E.A = lane_before(G.A, 1.0);
      E.B = lane_before(G.B, 0.0);
      if (wave == 1) E = aff_after(E, D0);
      double d = E.A * d_in + E.B;                 // d_t of this lane's latest step
      {
        const Aff DT = aff_after(D1, D0);
        d_in = DT.A * d_in + DT.B;
      }
      double dv[BS];
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        dv[j] = d;
        d = fa[j] * d + fb[j];                   // d_{t-1}
      }
      wave_block_store<BS>(stage, w0, tw, T, lane, true, dv);
    }
    d_first = d_in;
  }
  __syncthreads();

  KSTAMP(5);
  // ---- 6. forward: mean correction m_t = E(alpha_t | y) - E(alpha_t | y+) =
  // P0 d_{-1} + q sum_{s<t} d_s, the state draw alpha+_t + m_t, and the level
  // model's sufficient statistics (LocalLevelStateModel::observe_state)
  // ... and, in the same pass, the regression sufficient statistics given the
  // state (observe_data_given_state + NeRegSuf::add_mixture_data): residual
  // e_t = y_t - alpha_t on observed t (zero elsewhere) goes to array 1 of the
  // chain's block -- X'e for all chains is one GEMM after this kernel --
  // yty = e'e, n = #observed
  double lev_ss_part = 0.0, part_q = 0.0, part_n = 0.0;
  {
    double m_in = 0.0, st_in = 0.0;
    for (int t0 = 0; t0 < T; t0 += CS) {
      const int tl = t0 + WS * wave + BS * lane;
      // (all of the chunk's loads first: the stores below may alias them as far
      // as the compiler knows, and would serialise them)
      double dm[BS], al[BS], yv[BS];
      bool inr[BS], ob[BS];
      const int tw = t0 + WS * wave;   // the wave's first step
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int t = tl + j;
        inr[j] = t < T;
        ob[j] = (int)inr[j] & (int)(P.observed[inr[j] ? t : 0] != 0);
      }
      {
        double draw[BS], araw[BS], yraw[BS];
        wave_block_fetch<BS>(w0, (int64_t)tw - 1, (int64_t)T - 1, lane, 0.0, draw);   // d_{t-1}, 0 at t = 0 and past T
        wave_block_fetch<BS>(sal, tw, T, lane, 0.0, araw);
        wave_block_fetch<BS>(P.y, tw, T, lane, 0.0, yraw);
        wave_block_turn<BS>(stage, lane, false, draw, dm);
        wave_block_turn<BS>(stage, lane, false, araw, al);
        wave_block_turn<BS>(stage, lane, false, yraw, yv);
      }
      double mm[BS], st[BS];
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int t = tl + j;
        acc += !inr[j] ? 0.0 : ((t == 0) ? P.P0 * d_first : q * dm[j]);
        mm[j] = acc;
      }
      const double incl = wave_prefix_sum(acc);
      // the state's value at a wave's last step needs the other wave's sum first:
      // swap the sums, then (below) the states at the seams
      if (lane == WAVE - 1) s_x[wave][0] = incl;
      __syncthreads();
      const double m0tot = s_x[0][0], m1tot = s_x[1][0];
      const double base = m_in + (wave == 1 ? m0tot : 0.0) + (incl - acc);
      m_in += m0tot + m1tot;
      double last = 0.0;
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        st[j] = inr[j] ? al[j] + (mm[j] + base) : 0.0;
        if (inr[j]) last = st[j];
      }
      if (lane == WAVE - 1) s_x[wave][1] = last;
      __syncthreads();
      const double seam0 = s_x[0][1], seam1 = s_x[1][1];   // states at the two waves' last steps
      __syncthreads();
      // the state just before this lane's first step
      const double prev0 = lane_before(last, wave == 1 ? seam0 : st_in);
      st_in = seam1;
      double ev[BS];
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int t = tl + j;
        if (inr[j] && t > 0) {
          const double diff = st[j] - (j == 0 ? prev0 : st[j - 1]);
          lev_ss_part += diff * diff;
        }
        ev[j] = ob[j] ? yv[j] - st[j] : 0.0;
        if (ob[j]) { part_q += ev[j] * ev[j]; part_n += 1.0; }
      }
      wave_block_store<BS>(stage, sst, tw, T, lane, false, st);
      wave_block_store<BS>(stage, sF, tw, T, lane, false, ev);
    }
  }
  // the two waves' partial sums
  {
    const double a = wave_sum(lev_ss_part), b2 = wave_sum(part_q), c = wave_sum(part_n);
    if (lane == 0) { s_x[wave][0] = a; s_x[wave][1] = b2; s_x[wave][2] = c; }
    __syncthreads();
  }
  if (wave != 0) return;
  const double lev_ss = s_x[0][0] + s_x[1][0];
  const double lev_n = (double)(T - 1);
  KSTAMP(6);
  const double yty = s_x[0][1] + s_x[1][1];
  const double nobs = s_x[0][2] + s_x[1][2];
  if (lane == 0) {
    P.yty[chain] = yty;
    P.nobs[chain] = nobs;
    P.level_n[chain] = lev_n;
    P.level_sumsq[chain] = lev_ss;
#ifdef BA_KSTAMPS
    KSTAMP(7);
    if (chain == 0 && draw_level)
      printf("kalman phases (cycles): level %lld ystar %lld | normals: philox %lld fast %lld slow %lld tables %lld lookup %lld | forward %lld backward %lld correction %lld suf %lld\n",
             kph[0], kph[1], kph[2], kph[8], kph[9], kph[10], kph[11], kph[3] + kph[4], kph[5], kph[6], kph[7]);
#endif
    P.status[chain] = status;
  }
}

// The lane-major state draw (kalman_lm_device.h) as a kernel of its own: one chain per
// workgroup of 128.
__global__ __launch_bounds__(LM_THREADS, 2) void kalman_lm_kernel(SsParams P, int draw_level) {
  __shared__ KalmanLmLds lds;
  if ((int)blockIdx.x >= P.chain_count) return;
  (void)kalman_lm_body<false>(P, draw_level, (int)blockIdx.x + P.chain_first, lds);
}

// ... and the step that runs ahead of it (kalman_lm_device.h)
__global__ __launch_bounds__(128) void kalman_prepare_kernel(SsParams P, int draw_level) {
  __shared__ NormalsLds s_norm;
  if ((int)blockIdx.x >= P.chain_count) return;
  const int chain = (int)blockIdx.x + P.chain_first;
  kalman_prepare_body<false>(P, draw_level, chain, P.status[chain], s_norm);
}


// StateSpaceRegressionModel::simulate_forecast for every chain's current draw
// (StateSpaceRegressionModel.cpp:214-219, :256-278): state_i = state_{i-1} +
// N(0, sigma_level) starting from the final state, y_i = N(state_i, sigma_obs) +
// x_i'beta; the normals in the reference's order (state error, then
// observation) on the chain's forecast stream (id 5).  One wavefront per chain:
// the stream is read by the whole wave in lockstep, the lanes share x_i'beta.
__global__ __launch_bounds__(64) void ss_forecast_kernel(SsParams P, int horizon, const double *newX,
                                                         uint64_t *pos_forecast, double *out) {
  const int chain = (int)blockIdx.x + P.chain_first, lane = threadIdx.x;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) return;
  const int T = P.T, p = P.p;
  const double *beta = P.beta + (size_t)chain * p;
  const double sd_obs = sqrt(P.sigsq[chain]), sd_level = sqrt(P.level_sigsq[chain]);
  double state = P.scratch[(size_t)chain * P.scratch_stride + (size_t)SS_STATE_ARRAY * P.TP +
                           (P.lane_major ? lm_at(T - 1) : T - 1)];
  SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 5u}, pos_forecast[chain]};
  for (int i = 0; i < horizon; ++i) {
    state = state + d_rnorm(rng, 0.0, sd_level);
    const double obs = d_rnorm(rng, state, sd_obs);
    double part = 0.0;
    for (int j = lane; j < p; j += WAVE) part += newX[(size_t)j * horizon + i] * beta[j];
    const double pred = wave_sum(part);
    if (lane == 0) out[(size_t)chain * horizon + i] = obs + pred;
  }
  if (lane == 0) pos_forecast[chain] = rng.pos;
}

hipError_t launch_ss_forecast(hipStream_t stream, const SsParams &P, int horizon, const double *newX,
                              uint64_t *pos_forecast, double *out) {
  hipLaunchKernelGGL(ss_forecast_kernel, dim3(P.chain_count), dim3(WAVE), 0, stream, P, horizon, newX,
                     pos_forecast, out);
  return hipGetLastError();
}

hipError_t launch_xte_tiled(hipStream_t stream, const double *U, int64_t ldu, int R, const double *B, int64_t n,
                            int p, double *out, double *planes);

hipError_t launch_kalman_prepare(hipStream_t stream, const SsParams &P, int draw_level) {
  KtScope kt(stream, KT_KALMAN_PREPARE);
  hipLaunchKernelGGL(kalman_prepare_kernel, dim3(P.chain_count), dim3(2 * WAVE), 0, stream, P, draw_level);
  return hipGetLastError();
}

// the state draw itself ...
hipError_t launch_kalman_main(hipStream_t stream, const SsParams &P, int draw_level) {
  KtScope kt(stream, KT_KALMAN);
  if (P.lane_major)
    hipLaunchKernelGGL(kalman_lm_kernel, dim3(P.chain_count), dim3(LM_THREADS), 0, stream, P, draw_level);
  else
    hipLaunchKernelGGL(kalman_simsmooth_kernel, dim3(P.chain_count), dim3(2 * WAVE), 0,
                       stream, P, draw_level);
  return hipGetLastError();
}
// ... and xty[chain, j] = x_j' e_chain: residual series are array 1 of every chain's
// scratch block (zero where unobserved, and for a chain in error the previous
// sweep's -- its status stops it anyway)
// planes_only: the plane sum is left to the next SSVS launch (SsvsParams::xty_planes)
hipError_t launch_kalman_xte(hipStream_t stream, const SsParams &P, bool planes_only) {
  // (lane-major: residuals and Xt are the same permutation of time, zero past T)
  return launch_xte_tiled(stream, P.scratch + (size_t)P.chain_first * P.scratch_stride + P.TP, P.scratch_stride,
                          P.chain_count, P.lane_major ? P.Xt : P.X, (int64_t)P.TP, P.p,
                          planes_only ? nullptr : P.xty + (size_t)P.chain_first * P.p, P.xte_planes);
}

hipError_t launch_kalman_simsmooth(hipStream_t stream, const SsParams &P,
                                   int draw_level) {
  hipError_t err = launch_kalman_main(stream, P, draw_level);
  if (err != hipSuccess) return err;
  return launch_kalman_xte(stream, P, false);
}

}  // namespace boom_amd
