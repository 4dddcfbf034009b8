// Launch parameters of the Kalman filter / simulation smoother kernel
// (kalman_kernel.hip), shared with the host side (engine.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssvs_params.h"

namespace boom_amd {

// Structural state (ssm_kernel.hip): trend block (local level, or local linear
// trend) + optional seasonal block; state dimension m <= SSM_MAX.  Variance
// parameters are indexed 0 level, 1 slope, 2 seasonal.  An ArStateModel block
// (ar_lags coefficients, one error variance, its own sampler) may follow.
enum { SSM_MAX = 16 };
// a chain's ArModel sufficient statistics: xtx (lags x lags at leading dimension
// SSM_MAX) | xty | yty | n
enum { AR_SUF_XTY = SSM_MAX * SSM_MAX, AR_SUF_YTY = AR_SUF_XTY + SSM_MAX, AR_SUF_N = AR_SUF_YTY + 1,
       AR_SUF_STRIDE = AR_SUF_N + 1 };
struct SsmParams {
  int32_t m, trend, nseasons, s0;       // s0: first index of the seasonal block (-1: none)
  int32_t ar_lags, ar0;                 // the autoregression block: size (0: none), first index
  double prior_df[3], prior_ss[3], sigma_max[3];
  double ar_prior_df, ar_prior_ss, ar_sigma_max;
  double a0[SSM_MAX], P0[SSM_MAX];      // initial state mean, variance (diagonal)
  double *var_sigsq, *var_n, *var_ss;   // chains x 3
  uint64_t *pos_var;                    // chains x 3: streams 1, 6, 7
  double *ar_phi;                       // chains x SSM_MAX
  double *ar_sigsq;                     // chains
  double *ar_suf;                       // chains x AR_SUF_STRIDE
  uint64_t *pos_ar;                     // chains: stream 12 (the ArPosteriorSampler)
  // per chain: gains K (m x T) | state (m x T) | smoothed disturbances (4 x T) | normals
  double *work;
  int64_t work_stride;
};

struct SsParams {
  int32_t T, p, chains;
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
  SsmParams ssm;            // (ssm_kernel.hip only)
};

enum { SS_SCRATCH_ARRAYS = 9, SS_STATE_ARRAY = 4 };   // (arrays 5-6 and 7-8: the two normals buffers)
enum { LM_BS = 16, LM_THREADS = 128, LM_TP = LM_BS * LM_THREADS };
inline __host__ __device__ int lm_at(int t) { return (t % LM_BS) * LM_THREADS + t / LM_BS; }

}  // namespace boom_amd
