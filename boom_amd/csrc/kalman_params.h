// Launch parameters of the Kalman filter / simulation smoother kernel
// (kalman_kernel.hip), shared with the host side (engine.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssvs_params.h"

namespace boom_amd {

// Structural state (ssm_kernel.hip): BOOM's block-diagonal state -- any list of state
// models in the order they were added (StateSpaceModelBase::add_state): local level,
// local linear trend, seasonal (nseasons, season duration), autoregression, static intercept
// (StaticInterceptStateModel: on the device a local level whose variance is 0 and has no
// sampler -- nvar = 0, one variance SLOT that stays 0), trig (TrigStateModel: a 2 x 2 rotation
// per frequency, one variance for all 2 nfreq components).  State
// dimension m <= SSG_MAX_STATE (one lane per component), at most SSG_MAX_BLOCKS blocks,
// SSG_MAX_VAR variance parameters (two for a local linear trend, one otherwise; an
// autoregression block's error variance is one of them), SSG_MAX_AR autoregression
// blocks of at most AR_MAX lags.
enum { SSG_MAX_STATE = 64, SSG_MAX_BLOCKS = 8, SSG_MAX_VAR = 16, SSG_MAX_AR = 4, AR_MAX = 16 };
enum { SSG_LOCAL_LEVEL = 1, SSG_LOCAL_LINEAR_TREND = 2, SSG_SEASONAL = 3, SSG_AR = 4,
       SSG_STATIC_INTERCEPT = 5 /* (the C-ABI's number; stored as SSG_LOCAL_LEVEL with nvar = 0) */, SSG_TRIG = 6,
       // SemilocalLinearTrendStateModel: (level, slope, the slope's long-run mean mu); T = [[1, 1, 0],
       // [0, phi, 1 - phi], [0, 0, 1]]; its (phi, mu) and the slope's Ar1Suf take one of the chain's
       // autoregression slots (ar_phi[0 .. 1]; ar_suf[0 .. 5] = sumsq, sum, cross, n, first, last)
       SSG_SEMILOCAL = 7 };
// a chain's ArModel sufficient statistics (per autoregression block): xtx (lags x lags at
// leading dimension AR_MAX) | xty | yty | n
enum { AR_SUF_XTY = AR_MAX * AR_MAX, AR_SUF_YTY = AR_SUF_XTY + AR_MAX, AR_SUF_N = AR_SUF_YTY + 1,
       AR_SUF_STRIDE = AR_SUF_N + 1 };
struct SsgBlock {
  int32_t kind, first, dim, nvar;
  int32_t var0;        // index of the block's first variance parameter
  int32_t nseasons, duration, phase;   // seasonal: a new season starts at the times u with u % duration == phase
  int32_t lags, ar_index;              // autoregression: which of the chain's coefficient / suf slots
  int32_t sid[2];      // Philox sampler ids of the variance samplers (autoregression: its ArPosteriorSampler's)
  int32_t nfreq;       // trig: frequencies (dim = 2 nfreq)
  int32_t err0;        // index of the block's first state-error row (a trig block has dim of them, a trend two, the others one)
  int32_t sl_truncate, sl_positive;   // semilocal: NonzeroMeanAr1Sampler::force_stationary / force_ar1_positive
};
// the specification, in device memory (read through the scalar cache)
struct SsgSpec {
  int32_t m, nblocks, nvar, nar;
  int32_t ld;          // leading dimension of the state variance in LDS (odd)
  int32_t bl;          // steps per block of the passes (64 for m <= 16, ... 16 for m <= 64)
  int32_t nerr;        // rows of the state that carry state error (one per variance slot; a trig block: every component)
  int32_t pad;
  SsgBlock blk[SSG_MAX_BLOCKS];
  double prior_df[SSG_MAX_VAR], prior_ss[SSG_MAX_VAR], sigma_max[SSG_MAX_VAR];
  double a0[SSG_MAX_STATE], P0[SSG_MAX_STATE];   // initial state mean, variance (diagonal)
  // trig: the rotation [[c, s], [-s, c]] of the pair a component belongs to (by state component)
  double trig_c[SSG_MAX_STATE], trig_s[SSG_MAX_STATE];
  // semilocal (by autoregression slot): the slope's mean prior N(mu, sigma), its AR(1) coefficient's prior N(mu, sigma)
  double sl_prior[SSG_MAX_AR][4];
};
struct SsmParams {
  const SsgSpec *spec;                  // device copy
  int32_t m, nblocks, nvar, nar, ld, bl, nerr, pad;   // (the launch's copies of the scalars)
  // the block list is [local level | local linear trend] [+ seasonal of duration 1] [+ one
  // autoregression] with m <= 16: the shape-specialised kernel runs (ssm_template_kernel.hip);
  // tpl_trend = 0: the general kernel
  int32_t tpl_trend, tpl_nseasons, tpl_ar_lags;
  int32_t glob;                         // the list holds a trig or a semilocal-linear-trend block: the kernel instance that carries their code
  double *var_sigsq, *var_n, *var_ss;   // chains x SSG_MAX_VAR
  uint64_t *pos_var;                    // chains x SSG_MAX_VAR: the variance samplers' stream positions
  double *ar_phi;                       // chains x SSG_MAX_AR x AR_MAX
  double *ar_suf;                       // chains x SSG_MAX_AR x AR_SUF_STRIDE
  // per chain: gains K (m x T) | state (m x T) | smoothed disturbances (nerr x T) | normals
  double *work;
  int64_t work_stride;
};

struct SsParams {
  int32_t T, p, chains;
  int32_t slot_limit;     // > 0: uniforms a slot of the state stream serves before its spill stream (default: the stride)
  int32_t chain_first, chain_count;  // this launch: chains [chain_first, chain_first + chain_count)
  int64_t chain_offset;
  // shared data: StateSpaceRegressionModel(y, X, observed)
  const double *y;          // T
  const double *X;          // T x p column-major
  const uint8_t *observed;  // T
  // Lane-major copies for kalman_lm_kernel (local level, T <= LM_TP): step t of a series
  // sits at lm_at(t) = (t % 16) * 128 + t / 16 -- thread i of the chain's 128 owns steps
  // 16 i .. 16 i + 15 and its j-th step is element 128 j + i, so every per-step array
  // is read and written with plain coalesced accesses, no transposes.  Xt: p columns of
  // LM_TP (zero where unobserved -- never: as X -- and past T), yt: LM_TP, obs_mask[i]:
  // bit j = step 16 i + j observed.  lane_major = 1: the chains' scratch arrays
  // (residuals, state draw, normals) have pitch TP = LM_TP and that layout too.
  const double *Xt;
  const double *yt;
  const uint32_t *obs_mask;
  int32_t TP, lane_major;
  // regression parameters of every chain (written by the SSVS kernel)
  const uint8_t *gamma;     // chains x p
  const double *beta;       // chains x p
  const double *sigsq;      // chains
  // local level state model + its sampler
  double *level_sigsq;      // chains
  double *level_n;          // chains  (ZeroMeanGaussianModel suf)
  double *level_sumsq;      // chains
  double level_prior_df, level_prior_ss, level_sigma_max;
  double a0, P0;            // initial state mean / variance
  // RNG: stream 1 = level sampler, stream 2 = state imputation
  uint32_t seed_lo, seed_hi;
  uint64_t *pos_level, *pos_state;
  int32_t *status;
  const int32_t *only_ran;  // catch-up launches: skip chains whose entry is 0 (nullptr: all)
  // per-chain work arrays in HBM: v, F, K, v_sim, state, r, r_sim (T each)
  double *scratch;
  int64_t scratch_stride;   // SS_SCRATCH_ARRAYS x TP
  // per-chain regression sufficient statistics rebuilt by impute_state
  double *xty;              // chains x p
  double *yty;              // chains
  double *nobs;             // chains
  double *xte_planes;       // workspace of the X'e GEMM: xte_planes(T) x chains x p doubles
  // The part of a state draw that depends on nothing the same round's regression sweep
  // produces -- the level-variance draw and the sweep's standard normals -- is done ahead
  // by kalman_prepare_kernel on a second stream (see there).  zbuf = which of the chain's
  // two normals buffers this launch reads (main) or writes (prepare); prepared = 1: the
  // main kernel finds prep_n[zbuf][chain] normals there and level_sigsq drawn (0: not
  // prepared -- it does both itself; negative: the prepare step failed with that status)
  // and checks the count.  A chain the main kernel skips (parked by the sweep for want of
  // capacity) gets the prepare step rolled back from the prep_* copies, so that its
  // catch-up draws the same numbers.
  int32_t prepared, zbuf;
  int32_t *prep_n;             // 2 x chains
  uint64_t *prep_pos_state;    // 2 x chains: stream positions / level variance before the prepare step
  uint64_t *prep_pos_level;
  double *prep_level_sigsq;
  // the level variance THIS state draw used (the live value may already be the next round's:
  // the prepare step draws ahead); nullptr: not wanted.  Read by the look-ahead's record.
  double *level_used;
  SsmParams ssm;            // (ssm_kernel.hip only)
};

// The round kernel (ss_round_kernel.hip): every chain's workgroup loops over the rounds of a
// call by itself; the chains meet only in the X'e tiles -- up to SS_ROUND_TILE chains in the
// order they arrive.  Per round r: ticket[r] = the next place (tile * SS_ROUND_TILE + slot; a
// tile closed early moves it to the next tile's first place), members = the chain index in
// every place (-1: not yet written), sizes[tile] = members of a tile closed early (0: not).
// At most one tile per chain and round.  The host zeroes them (members: -1) before every launch.
enum { SS_ROUND_TILE = 16, SS_ROUND_MAX_ROUNDS = 64 };
struct SsRoundParams {
  int32_t rounds;           // rounds of this launch (<= SS_ROUND_MAX_ROUNDS)
  int32_t close_ticks;      // 100 MHz ticks a tile's first member waits for company
  int32_t *ticket;          // [rounds]
  int32_t *sizes;           // [rounds][chain_count]
  int32_t *members;         // [rounds][chain_count * SS_ROUND_TILE + 2 * SS_ROUND_TILE]
  double *planes;           // SS_ROUND_TILE x chains x p: a chain's partial products, by row of the series
  // the look-ahead's record of every round's draw (engine.hip; rgamma == nullptr: none):
  // chain c's row of round r is (rec_slot * chains + c) * rec_len + rec_first + r
  uint8_t *rgamma;
  double *rbeta, *rsig, *rvar, *rstate;
  const int32_t *reg_of_chain;      // chains: index among the chains whose state path is recorded, or -1
  int32_t rec_slot, rec_len, rec_first, nreg;
  int32_t *debug;           // diagnostic (ba_ss_set_tuning(e, 6) in a debugging session): 16 x 16 words, else nullptr
  int32_t debug_seq;        // diagnostic build: a number per launch
  double *stamps;           // diagnostic build (-DBA_RSTAMPS): chains x 2 waves x 8 phases, 100 MHz ticks; else unused
};

enum { SS_SCRATCH_ARRAYS = 9, SS_STATE_ARRAY = 4 };   // (arrays 5-6 and 7-8: the two normals buffers)
enum { LM_BS = 16, LM_THREADS = 128, LM_TP = LM_BS * LM_THREADS };
inline __host__ __device__ int lm_at(int t) { return (t % LM_BS) * LM_THREADS + t / LM_BS; }

}  // namespace boom_amd
