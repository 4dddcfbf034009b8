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

#include "device_rng.h"
#include "kalman_params.h"
#include "stream_normals.h"

namespace boom_amd {

namespace {

constexpr int WAVE = 64;

// diagnostic build (-DBA_KSTAMPS): chain 0 prints its cycles per phase
#ifdef BA_KSTAMPS
#define KSTAMP(i) do { const long long t_ = (long long)__builtin_readcyclecounter(); kph[i] += t_ - klast; klast = t_; } while (0)
#else
#define KSTAMP(i) do { } while (0)
#endif

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

typedef __attribute__((address_space(3))) double AS_LDS_F64;

// x -> A x + B
struct Aff { double A, B; };
__device__ __forceinline__ Aff aff_after(const Aff &later, const Aff &earlier) {
  Aff r;
  r.A = later.A * earlier.A;
  r.B = later.A * earlier.B + later.B;
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Aff aff_dpp(const Aff &f) {
  Aff r;
  r.A = dpp_f64<CTRL, ROW_MASK>(f.A, 1.0);  // lanes without a source get the identity
  r.B = dpp_f64<CTRL, ROW_MASK>(f.B, 0.0);
  return r;
}
// inclusive scan over the wave: lane i ends with f_i o f_{i-1} o ... o f_0
__device__ __forceinline__ Aff wave_scan(Aff f) {
  f = aff_after(f, aff_dpp<0x111, 0xf>(f));  // row_shr:1
  f = aff_after(f, aff_dpp<0x112, 0xf>(f));  // row_shr:2
  f = aff_after(f, aff_dpp<0x114, 0xf>(f));  // row_shr:4
  f = aff_after(f, aff_dpp<0x118, 0xf>(f));  // row_shr:8
  f = aff_after(f, aff_dpp<0x142, 0xa>(f));  // row_bcast:15 into rows 1, 3
  f = aff_after(f, aff_dpp<0x143, 0xc>(f));  // row_bcast:31 into rows 2, 3
  return f;
}
// x -> (a x + b) / (c x + d)
struct Mob { double a, b, c, d; };
__device__ __forceinline__ Mob mob_after(const Mob &l, const Mob &e) {
  Mob r;
  r.a = l.a * e.a + l.b * e.c;
  r.b = l.a * e.b + l.b * e.d;
  r.c = l.c * e.a + l.d * e.c;
  r.d = l.c * e.b + l.d * e.d;
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Mob mob_dpp(const Mob &f) {
  Mob r;
  r.a = dpp_f64<CTRL, ROW_MASK>(f.a, 1.0);
  r.b = dpp_f64<CTRL, ROW_MASK>(f.b, 0.0);
  r.c = dpp_f64<CTRL, ROW_MASK>(f.c, 0.0);
  r.d = dpp_f64<CTRL, ROW_MASK>(f.d, 1.0);
  return r;
}
__device__ __forceinline__ Mob wave_scan(Mob f) {
  f = mob_after(f, mob_dpp<0x111, 0xf>(f));
  f = mob_after(f, mob_dpp<0x112, 0xf>(f));
  f = mob_after(f, mob_dpp<0x114, 0xf>(f));
  f = mob_after(f, mob_dpp<0x118, 0xf>(f));
  f = mob_after(f, mob_dpp<0x142, 0xa>(f));
  f = mob_after(f, mob_dpp<0x143, 0xc>(f));
  return f;
}
// inclusive prefix sum, lane order
__device__ __forceinline__ double wave_prefix_sum(double x) {
  x += dpp_f64<0x111, 0xf>(x, 0.0);
  x += dpp_f64<0x112, 0xf>(x, 0.0);
  x += dpp_f64<0x114, 0xf>(x, 0.0);
  x += dpp_f64<0x118, 0xf>(x, 0.0);
  x += dpp_f64<0x142, 0xa>(x, 0.0);
  x += dpp_f64<0x143, 0xc>(x, 0.0);
  return x;
}
// the value of lane - 1 (lane 0 gets `first`): wave_shr:1
__device__ __forceinline__ double lane_before(double x, double first) {
  return dpp_f64<0x138, 0xf>(x, first);
}

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
  const bool prepared = P.prepared != 0 && *prep_n_slot > 0;
  if (P.prepared != 0 && *prep_n_slot < 0 && P.status[chain] == CHAIN_OK) {   // the prepare step failed
    if (threadIdx.x == 0) { P.status[chain] = -*prep_n_slot; *prep_n_slot = 0; }
    return;
  }
  if (P.status[chain] != CHAIN_OK || (P.only_ran && P.only_ran[chain] == 0)) {
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
  if (!prepared || *prep_n_slot != N) {
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
        obs[j] = in[j] && P.observed[in[j] ? t : 0] != 0;
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
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const int t = tw + i * WAVE + lane;
            yraw[i] = (t < T) ? P.y[t] - pred[i] : 0.0;
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
              const bool obx = inx && P.observed[inx ? t : 0] != 0;
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
        ob[j] = inr[j] && P.observed[inr[j] ? t : 0] != 0;
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

// ---------------------------------------------------------------------------------
// The same state draw for a series of at most LM_TP = 2048 steps, in the LANE-MAJOR
// layout (kalman_params.h): thread i of the chain's 128 owns steps 16 i .. 16 i + 15
// through ALL passes, so what the passes hand to each other -- K_t, (v_t - v+_t) / F_t,
// alpha+_t, d_{t-1} -- never leaves its registers, and what comes from or goes to memory
// (X columns, y, the normals, the residuals, the state) is element 128 j + i for its
// j-th step: coalesced as it stands.  No work arrays, no transposes, one chunk.  The
// backward pass keeps the ownership and scans the lanes' composites in reverse order
// (wave 1 before wave 0, lane 63 before lane 0).
// Measured where the kernel above spent its time at T = 2000 (chain 0's cycle stamps):
// y* 33 k cycles of 86 k (one memory round trip per included variable and chunk, one
// after the other), the block transposes 11 k, the backward pass 12 k, the correction
// pass 22 k.  Here: the variables' columns two at a time, everything else as above.
struct LmSlots {   // slot s of the normals array: row s / 128 = 2 j + kind, thread s % 128
  int T, nfirst, nper, dI, dL, dH;
  __device__ __forceinline__ int count() const { return 2 * LM_TP; }
  __device__ __forceinline__ int draw(int s) const {
    const int row = s >> 7, kind = row & 1;
    const int t = LM_BS * (s & (LM_THREADS - 1)) + (row >> 1);
    if (t >= T) return -1;
    if (t == 0) return kind == 0 ? (dI ? 0 : -1) : (dH ? dI : -1);
    if (kind == 0) return dL ? nfirst + (t - 1) * nper : -1;
    return dH ? nfirst + (t - 1) * nper + dL : -1;
  }
};

__global__ __launch_bounds__(LM_THREADS, 2) void kalman_lm_kernel(SsParams P, int draw_level) {
  constexpr int BS = LM_BS, NT = LM_THREADS;
  constexpr int LM_VB = 5;                 // variables' columns in flight together in the y* pass
  __shared__ NormalsLds s_norm;            // (only a chain whose normals were not prepared uses it)
  __shared__ double s_x[2][8];             // the two waves' scan totals
  __shared__ uint32_t s_mask[NT];          // (H == 0 only: the threads' observed masks)
  if ((int)blockIdx.x >= P.chain_count) return;
  const int chain = (int)blockIdx.x + P.chain_first, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (the chain's scalars first, all in flight together, then the decisions)
  int32_t *prep_n_slot = P.prep_n + (size_t)P.zbuf * P.chains + chain;
  const int prep_n = *prep_n_slot, status_in = P.status[chain];
  const int ran = P.only_ran ? P.only_ran[chain] : 1;
  double level_sigsq = P.level_sigsq[chain];
  const double sigsq_obs = P.sigsq[chain];
  const bool prepared = P.prepared != 0 && prep_n > 0;
  if (P.prepared != 0 && prep_n < 0 && status_in == CHAIN_OK) {   // the prepare step failed
    if (tid == 0) { P.status[chain] = -prep_n; *prep_n_slot = 0; }
    return;
  }
  if (status_in != CHAIN_OK || ran == 0) {
    if (prepared && tid == 0) {
      P.pos_state[chain] = P.prep_pos_state[(size_t)P.zbuf * P.chains + chain];
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
  if (draw_level && !prepared) {
    SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 1u}, P.pos_level[chain]};
    int bad = 0;
    const double DF = P.level_n[chain] + P.level_prior_df;
    const double SS = P.level_sumsq[chain] + P.level_prior_ss;
    level_sigsq = d_draw_variance(rng, DF, SS, P.level_sigma_max, &bad);
    if (bad) status = CHAIN_RNG_BRANCH;
    __syncthreads();   // (every thread has read the position it draws from)
    if (tid == 0) {
      P.pos_level[chain] = rng.pos;
      P.level_sigsq[chain] = level_sigsq;
    }
  }
  if (status != CHAIN_OK) {
    if (tid == 0) P.status[chain] = status;
    return;
  }
  if (P.level_used && tid == 0) P.level_used[chain] = level_sigsq;
  const double *beta = P.beta + (size_t)chain * p;
  double *base = P.scratch + (size_t)chain * P.scratch_stride;
  double *sF = base + (size_t)P.TP;                         // residuals (input of the X'e GEMM)
  double *sst = base + (size_t)SS_STATE_ARRAY * P.TP;       // the state draw
  double *szz = base + (size_t)(5 + 2 * P.zbuf) * P.TP;     // the sweep's normals, LmSlots layout (2 TP; two buffers)
  const double q = level_sigsq, level_sigma = sqrt(level_sigsq);
  const double H = sigsq_obs, sqrtH = sqrt(H), sd0 = sqrt(P.P0);
  const int dI = (sd0 != 0.0), dL = (level_sigma != 0.0), dH = (sqrtH != 0.0);
  const int nfirst = dI + dH, nper = dL + dH;
  const int N = nfirst + (T - 1) * nper;
  KSTAMP(0);
  if (!prepared || prep_n != N) {   // (another count: see kalman_simsmooth_kernel)
    const uint64_t bpos0 = prepared ? P.prep_pos_state[(size_t)P.zbuf * P.chains + chain] : P.pos_state[chain];
    status = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, bpos0, N, szz,
                            &P.pos_state[chain], LmSlots{T, nfirst, nper, dI, dL, dH}, ss_slot_serve(P));
    if (status != CHAIN_OK) {
      if (tid == 0) P.status[chain] = status;
      return;
    }
    __syncthreads();
  }
  if (prepared && tid == 0) *prep_n_slot = 0;   // consumed
  KSTAMP(1);

  // the thread's steps: tl + j, j < 16; bit j of inm: inside the series, of obm: observed
  const int tl = BS * tid;
  const uint32_t inm = (tl + BS <= T) ? 0xffffu : (tl < T ? (1u << (T - tl)) - 1u : 0u);
  const uint32_t obm = P.obs_mask[tid] & inm;
#define LM_IN(j) (((inm >> (j)) & 1u) != 0u)
#define LM_OB(j) (((obm >> (j)) & 1u) != 0u)

  // ---- 1. y*_t = y_t - x_t'beta: the included variables' columns LM_VB at a time (their
  // loads in flight together); products accumulate in variable order, as GlmCoefs::predict
  double ys[BS];
  {
    double pred[BS], yv[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) { pred[j] = 0.0; yv[j] = P.yt[j * NT + tid]; }
    for (int vb = 0; vb < p; vb += WAVE) {
      const int jv = vb + lane;
      const double bj = (jv < p) ? beta[jv] : 0.0;
      unsigned long long mk = __ballot(bj != 0.0);
      while (mk) {
        // up to LM_VB variables of the batch (absent ones repeat the first with a zero
        // coefficient that is never added)
        int l[LM_VB];
        int cnt = 0;
#pragma unroll
        for (int v = 0; v < LM_VB; ++v) {
          const bool have = mk != 0;
          l[v] = have ? __ffsll((long long)mk) - 1 : l[0];
          if (have) { mk &= mk - 1; ++cnt; }
        }
        double b[LM_VB];
        const double *c[LM_VB];
#pragma unroll
        for (int v = 0; v < LM_VB; ++v) {
          b[v] = bcast_u(bj, l[v]);
          c[v] = P.Xt + (size_t)(vb + l[v]) * LM_TP + tid;
        }
        double x[LM_VB][BS];
#pragma unroll
        for (int v = 0; v < LM_VB; ++v)
#pragma unroll
          for (int j = 0; j < BS; ++j) x[v][j] = c[v][j * NT];
#pragma unroll
        for (int v = 0; v < LM_VB; ++v) {
          if (v < cnt) {
#pragma unroll
            for (int j = 0; j < BS; ++j) pred[j] += x[v][j] * b[v];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BS; ++j) ys[j] = LM_IN(j) ? yv[j] - pred[j] : 0.0;
  }
  KSTAMP(2);
  __builtin_amdgcn_sched_barrier(0);   // (registers: y* is done before the normals come in)
  // ---- 2. the sweep's normals: state error (initial state at t = 0) and observation error
  double zL[BS], zH[BS];
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    zL[j] = szz[(2 * j) * NT + tid];
    zH[j] = szz[(2 * j + 1) * NT + tid];
  }
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    const bool first = (tl + j == 0);
    zL[j] = (LM_IN(j) && (first ? dI : dL)) ? zL[j] : 0.0;
    zH[j] = (LM_IN(j) && dH) ? zH[j] : 0.0;
  }
  KSTAMP(3);

  // ---- 3 + 4. forward pass (see kalman_simsmooth_kernel): variances as a scan of Moebius
  // maps, alpha+ as a prefix sum, the filter on w = y* - y+ as a scan of affine maps
  double K[BS], al[BS], ef[BS];
  {
    const bool moebius = H > 0.0;
    const double r = moebius ? q / H : 0.0, s1 = 1.0 / (1.0 + r);
    const double u_in = moebius ? P.P0 / H : P.P0;
    Mob M;
    M.a = 1.0; M.b = 0.0; M.c = 0.0; M.d = 1.0;
    double asum = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      Mob mt;
      mt.a = 1.0;
      mt.b = LM_OB(j) ? r * s1 : (LM_IN(j) ? r : 0.0);
      mt.c = LM_OB(j) ? s1 : 0.0;
      mt.d = LM_OB(j) ? s1 : 1.0;
      M = mob_after(mt, M);
      asum += !LM_IN(j) ? 0.0 : ((tl + j == 0) ? P.a0 + sd0 * zL[j] : level_sigma * zL[j]);
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
    if (!moebius) s_mask[tid] = obm | (inm << 16);
    __syncthreads();
    Mob G0;
    G0.a = s_x[0][0]; G0.b = s_x[0][1]; G0.c = s_x[0][2]; G0.d = s_x[0][3];
    const double a0tot = s_x[0][4];
    double w[BS];
    {
      const double carry = (wave == 1 ? a0tot : 0.0) + (aincl - asum);
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        al[j] += carry;
        w[j] = LM_IN(j) ? ys[j] - (al[j] + sqrtH * zH[j]) : 0.0;
      }
    }
    double Fv[BS];
    if (moebius) {
      Mob E;
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
        K[j] = LM_OB(j) ? kk : 0.0;
        u = LM_OB(j) ? kk + r : (LM_IN(j) ? u + r : u);
      }
    } else {
      // H == 0: the plain recursion, thread after thread (every thread runs all of it and
      // keeps the values of its own steps)
      double Pv = u_in;
      for (int th = 0; th < NT; ++th) {
        const uint32_t mm = s_mask[th];
        double Pl = Pv;
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const bool inx = ((mm >> (16 + j)) & 1u) != 0u, obx = ((mm >> j) & 1u) != 0u;
          const double PZ = Pl, Fi = PZ + H;
          const double Ki = obx ? PZ / Fi : 0.0;
          if (th == tid) { Fv[j] = Fi; K[j] = Ki; }
          if (obx) Pl = Pl + (-1.0) * PZ * Ki;
          if (inx) Pl = Pl + q;
        }
        Pv = Pl;
      }
    }
    bool badF = false;
#pragma unroll
    for (int j = 0; j < BS; ++j) badF = badF || (LM_IN(j) && !(Fv[j] > 0.0));
    const bool anybad = __any(badF) != 0;
    Aff C;
    C.A = 1.0; C.B = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
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
    Aff D0;
    D0.A = s_x[0][0]; D0.B = s_x[0][1];
    const bool bad_any = (s_x[0][2] != 0.0) || (s_x[1][2] != 0.0);
    __syncthreads();
    if (bad_any) {
      if (tid == 0) P.status[chain] = CHAIN_FORECAST_VARIANCE;
      return;
    }
    Aff Ed;
    Ed.A = lane_before(Gd.A, 1.0);
    Ed.B = lane_before(Gd.B, 0.0);
    if (wave == 1) Ed = aff_after(Ed, D0);
    double delta = Ed.B;   // (delta_0 = 0 goes in)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      ef[j] = LM_OB(j) ? (w[j] - delta) / Fv[j] : 0.0;   // (v_t - v+_t) / F_t
      delta = (1.0 - K[j]) * delta + K[j] * w[j];
    }
  }
  KSTAMP(4);

  // ---- 5. backward: d_{t-1} = e_t / F_t + (1 - K_t) d_t, d_{T-1} = 0, over the same
  // ownership: the thread's own steps latest first, d_{lo-1} = (f_lo o ... o f_hi)(d_hi),
  // then the composites in scan order = time downwards
  double dm[BS];   // d_{t-1} of the thread's own steps
  {
    Aff C;
    C.A = 1.0; C.B = 0.0;
#pragma unroll
    for (int j = BS - 1; j >= 0; --j) {
      Aff f;
      f.A = 1.0 - K[j];   // (1 at steps past T: K is 0 there, and e / F too)
      f.B = ef[j];
      C = aff_after(f, C);
    }
    Aff Cr;
    Cr.A = __shfl(C.A, WAVE - 1 - lane);
    Cr.B = __shfl(C.B, WAVE - 1 - lane);
    const Aff G = wave_scan(Cr);
    if (lane == WAVE - 1) { s_x[wave][0] = G.A; s_x[wave][1] = G.B; }
    __syncthreads();
    Aff D1;
    D1.A = s_x[1][0]; D1.B = s_x[1][1];
    Aff E;
    E.A = lane_before(G.A, 1.0);
    E.B = lane_before(G.B, 0.0);
    if (wave == 0) E = aff_after(E, D1);
    double d = __shfl(E.B, WAVE - 1 - lane);   // d at the thread's latest step (0 goes in at T - 1)
#pragma unroll
    for (int j = BS - 1; j >= 0; --j) {
      d = (1.0 - K[j]) * d + ef[j];
      dm[j] = d;
    }
    __syncthreads();
  }
  KSTAMP(5);

  // ---- 6. forward: mean correction m_t = P0 d_{-1} + q sum_{s<t} d_s, the state draw
  // alpha+_t + m_t, the level model's sufficient statistics, and the regression's given
  // the state: residual e_t = y_t - alpha_t at observed t (0 elsewhere), e'e, #observed
  // (nothing is stored before the last exchange: a barrier waits for the stores in flight)
  {
    // (loaded here, not under the forward pass: a barrier there would wait for it)
    double yv[BS];
    {
      // (four steps per address register, made HERE: the sixteen addresses computed at the
      // top of the kernel sat in scratch memory until now -- ten serial reloads)
      const double *yb = P.yt + tid;
#pragma unroll
      for (int g = 0; g < BS / 4; ++g) {
        const double *q = yb + (size_t)4 * g * NT;
        asm volatile("" : "+v"(q));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) yv[4 * g + jj] = q[jj * NT];
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      acc += !LM_IN(j) ? 0.0 : ((tl + j == 0) ? P.P0 * dm[j] : q * dm[j]);
      dm[j] = acc;   // m_t less the carry
    }
    const double incl = wave_prefix_sum(acc);
    if (lane == WAVE - 1) s_x[wave][0] = incl;
    __syncthreads();
    const double carry = (wave == 1 ? s_x[0][0] : 0.0) + (incl - acc);
    double last = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      al[j] = LM_IN(j) ? al[j] + (dm[j] + carry) : 0.0;   // the state draw
      if (LM_IN(j)) last = al[j];
    }
    if (lane == WAVE - 1) s_x[wave][1] = last;
    __syncthreads();
    const double prev0 = lane_before(last, wave == 1 ? s_x[0][1] : 0.0);   // the state before the thread's first step
    double lev_ss_part = 0.0, part_q = 0.0, part_n = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      if (LM_IN(j) && tl + j > 0) {
        const double diff = al[j] - (j == 0 ? prev0 : al[j - 1]);
        lev_ss_part += diff * diff;
      }
      yv[j] = LM_OB(j) ? yv[j] - al[j] : 0.0;   // the residual
      if (LM_OB(j)) { part_q += yv[j] * yv[j]; part_n += 1.0; }
    }
    const double a = wave_sum(lev_ss_part), b2 = wave_sum(part_q), c = wave_sum(part_n);
    if (lane == 0) { s_x[wave][2] = a; s_x[wave][3] = b2; s_x[wave][4] = c; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      sst[j * NT + tid] = al[j];
      sF[j * NT + tid] = yv[j];
    }
  }
#undef LM_IN
#undef LM_OB
  KSTAMP(6);
  if (tid == 0) {
    P.yty[chain] = s_x[0][3] + s_x[1][3];
    P.nobs[chain] = s_x[0][4] + s_x[1][4];
    P.level_n[chain] = (double)(T - 1);
    P.level_sumsq[chain] = s_x[0][2] + s_x[1][2];
#ifdef BA_KSTAMPS
    KSTAMP(7);
    if (chain == 0 && draw_level)
      printf("kalman lane-major phases (cycles): start %lld normals %lld ystar %lld z %lld forward %lld backward %lld correction %lld suf %lld\n",
             kph[0], kph[1], kph[2], kph[3], kph[4], kph[5], kph[6], kph[7]);
#endif
    P.status[chain] = status;
  }
}

// The two pieces of a state draw that do not depend on the same round's regression
// sweep, done ahead of it on the engine's second stream (SsParams::prepared):
// ZeroMeanGaussianConjSampler::draw for the level variance (its own stream, the level
// model's sufficient statistics of the previous state draw) and the standard normals of
// simulate_forward (stream positions only), into the normals buffer P.zbuf.  The step
// for round r + 1 goes out behind round r's state draw and runs beside round r's X'e
// GEMM, plane sum and the start of round r + 1's SSVS launch (whose wavefronts raise
// their issue priority: the generator is bound by 32-bit multiplies, two such wavefronts
// to a SIMD took 9 us from a launch of one-wave chains).  Beside the state draw itself
// it cannot run: that kernel's wavefronts fill the register files (measured: 64 us
// instead of 30 when the prepare step's workgroups got there first).
// Which draws exist (a zero variance draws nothing, Bmath/rnorm.cpp:63-64) is taken from
// the level variance just drawn and a positive observation variance; the main kernel
// checks the count.  A failure is handed over as a negative count.
__global__ __launch_bounds__(128) void kalman_prepare_kernel(SsParams P, int draw_level) {
  __shared__ NormalsLds s_norm;
  if ((int)blockIdx.x >= P.chain_count) return;
  const int chain = (int)blockIdx.x + P.chain_first;
  if (P.status[chain] != CHAIN_OK) return;
  const int T = P.T;
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  const uint64_t pos_level0 = P.pos_level[chain], pos_state0 = P.pos_state[chain];
  const double level0 = P.level_sigsq[chain];
  double level_sigsq = level0;
  int status = CHAIN_OK;
  if (draw_level) {
    SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 1u}, pos_level0};
    int bad = 0;
    const double DF = P.level_n[chain] + P.level_prior_df;
    const double SS = P.level_sumsq[chain] + P.level_prior_ss;
    level_sigsq = d_draw_variance(rng, DF, SS, P.level_sigma_max, &bad);
    if (bad) status = CHAIN_RNG_BRANCH;
    __syncthreads();   // (every thread has read the statistics and the position it draws from)
    if (threadIdx.x == 0) {
      P.pos_level[chain] = rng.pos;
      P.level_sigsq[chain] = level_sigsq;
    }
  }
  const int dI = (sqrt(P.P0) != 0.0), dL = (sqrt(level_sigsq) != 0.0), dH = 1;
  const int N = (dI + dH) + (T - 1) * (dL + dH);
  double *szz = P.scratch + (size_t)chain * P.scratch_stride + (size_t)(5 + 2 * P.zbuf) * P.TP;
  if (status == CHAIN_OK) {   // (uniform: every thread made the same draw)
    if (P.lane_major)
      status = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, pos_state0, N, szz,
                              &P.pos_state[chain], LmSlots{T, dI + dH, dL + dH, dI, dL, dH}, ss_slot_serve(P));
    else
      status = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, pos_state0, N, szz,
                              &P.pos_state[chain], ss_slot_serve(P));
  }
  if (threadIdx.x == 0) {
    const size_t slot = (size_t)P.zbuf * P.chains + chain;
    P.prep_n[slot] = (status == CHAIN_OK) ? N : -status;
    P.prep_pos_state[slot] = pos_state0;
    P.prep_pos_level[slot] = pos_level0;
    P.prep_level_sigsq[slot] = level0;
  }
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
