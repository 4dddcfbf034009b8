/* TEST INFRASTRUCTURE -- CPU oracle, not part of the shipped product.
 *
 * Clean-room C restatement of the BOOM hot path: the
 * BregVsSampler SSVS Gibbs sweep and the scalar Kalman filter / Durbin-Koopman
 * simulation smoother behind StateSpacePosteriorSampler, written against flat
 * arrays with the same control flow and RNG consumption order as the
 * reference.  Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  Nothing under boom_amd/ links,
 * imports or calls it.
 *
 * Parity status: PINNED.  With the MT19937-64 uniform source the restatement
 * reproduces the compiled, unmodified reference (oracle/_ref/libboomref.so)
 * on the same seeds: inclusion indicators identical, continuous draws to
 * <= 1e-9 relative (Eigen's vectorised reductions round differently from the
 * plain loops here).  The fixtures in tests/golden/ were produced by that
 * reference build (tests/golden/make_golden.py) and are re-checked on every
 * test run, including on the GPU box where the reference itself is absent.
 */
#ifndef BOOM_ORACLE_H
#define BOOM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ RNG */
/* Uniform source with BOOM::RNG semantics (distributions/rng.hpp:27-54): one
 * double in [0,1) per call.  Two interchangeable engines:
 *   BO_RNG_MT      std::mt19937_64 + libstdc++ uniform_real_distribution, the
 *                  reference's own engine -- used to pin this file against it.
 *   BO_RNG_PHILOX  Philox4x32-10 (Salmon et al., SC'11) counter stream, the
 *                  engine of the HIP kernels: uniform number i of stream
 *                  (seed, chain, stream) comes from block i>>1, 64-bit half
 *                  i&1, u = (x >> 11) * 2^-53.
 */
enum { BO_RNG_MT = 0, BO_RNG_PHILOX = 1 };

typedef struct bo_rng {
  int kind;
  /* mt19937_64 */
  uint64_t mt[312];
  int mti;
  /* philox */
  uint64_t seed;
  uint32_t chain;
  uint32_t stream;
  uint64_t pos; /* index of the next uniform */
  /* The chain's state stream (stream 2: the normals of simulate_forward): normal
   * number `slot` reads its uniforms from the fixed position slot * slot_stride, as
   * the device does (stream_normals.h) -- every draw independent of the others.
   * 0 for every other stream, and the MT engine reads in sequence as the reference. */
  uint64_t slot_stride, slot;
  /* A slot of a substream (the state stream's normals, the imputers' observations): the
   * first position that is not the slot's own, and where its SPILL stream starts (stream
   * id | BO_SPILL_STREAM_BIT, position index << BO_SPILL_SHIFT) -- a draw that needs more
   * uniforms than the slot serves goes on there (device_rng.h does the same).  limit 0:
   * no slot. */
  uint64_t limit, spill;
} bo_rng;
#define BO_STATE_SLOT_STRIDE 256
#define BO_SPILL_STREAM_BIT 0x80000000u
#define BO_SPILL_SHIFT 20
/* slot `index` of a stream with `stride` positions per draw */
void bo_rng_slot(bo_rng *r, uint64_t index, uint64_t stride);
/* tests: a slot serves only `uniforms` numbers before its spill stream (0: the whole
 * stride; the device's ba_set_slot_limit is the same switch) */
void bo_set_slot_limit(int uniforms);

void bo_rng_seed_mt(bo_rng *r, uint64_t seed);
void bo_rng_seed_philox(bo_rng *r, uint64_t seed, uint32_t chain,
                        uint32_t stream, uint64_t pos);
double bo_unif(bo_rng *r);
/* BOOM::seed_rng, distributions/rng.cpp:39-47 */
uint64_t bo_seed_rng(bo_rng *r);
void bo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                      uint32_t out[4]);

double bo_runif(bo_rng *r, double a, double b);
int bo_random_int(bo_rng *r, int lo, int hi);
void bo_shuffle(bo_rng *r, int *v, int n);
double bo_norm_rand(bo_rng *r);
double bo_rnorm(bo_rng *r, double mu, double sigma);
double bo_exp_rand(bo_rng *r);
/* BOOM::rgamma_mt(rng, a, b): shape a, RATE b.  *status != 0 on a branch the
 * oracle does not restate (a < 0.3). */
double bo_rgamma(bo_rng *r, double a, double b, int *status);
double bo_rtrun_gamma(bo_rng *r, double a, double b, double cut, int *status);
int bo_rmulti(bo_rng *r, const double *prob, int n, int *status);

/* ---------------------------------------------------------------- LinAlg */
/* All matrices column-major, full storage (LinAlg/Matrix.hpp:429). */
int bo_chol(int n, const double *A, double *L);      /* 1 = positive definite */
double bo_spd_logdet(int n, const double *A, int *ok);
int bo_spd_solve(int n, const double *A, const double *rhs, double *x);
double bo_spd_mdist(int n, const double *A, const double *x);

/* NeRegSuf(X, y), Models/Glm/RegressionModel.cpp:309-328 */
void bo_neregsuf(int n, int p, const double *X, const double *y, double *xtx,
                 double *xty, double *yty, double *sumy, double *xsum);

/* ------------------------------------------------------------------ SSVS */
enum {
  BO_OK = 0,
  BO_ERR_NOT_PD = 1,        /* draw_beta failed > 10 times */
  BO_ERR_NEGATIVE_SS = 2,   /* set_reg_post_params: SS < 0 or non-finite */
  BO_ERR_ILLEGAL_START = 3, /* draw_model_indicators: no legal configuration */
  BO_ERR_UNSUPPORTED_RNG_BRANCH = 4,
  BO_ERR_FORECAST_VARIANCE = 5,
  BO_ERR_INVALID = 6        /* an argument or a state the reference reports as an error */
};

typedef struct bo_ssvs bo_ssvs;

/* Sufficient statistics + raw priors (BregVsSampler ctor #5 semantics,
 * BregVsSampler.cpp:180-194).  prior_df / sigma_guess parameterise
 * ChisqModel(df, sigma_guess) (ChisqModel.cpp:56-57). */
bo_ssvs *bo_ssvs_create(int p, const double *xtx, const double *xty,
                        double yty, double n, double sumy, const double *xsum,
                        const double *prior_mean, const double *ominv,
                        double prior_df, double sigma_guess, const double *pi);
void bo_ssvs_set_priors(bo_ssvs *s, const double *prior_mean, const double *ominv,
                        double prior_df, double sigma_guess, const double *pi);
void bo_ssvs_destroy(bo_ssvs *s);
void bo_ssvs_set_options(bo_ssvs *s, int64_t max_model_size,
                         double sigma_upper_limit, double swap_threshold,
                         int max_flips, int draw_beta, int draw_sigma);
void bo_ssvs_set_state(bo_ssvs *s, const uint8_t *gamma, const double *beta,
                       double sigsq);
void bo_ssvs_get_state(const bo_ssvs *s, uint8_t *gamma, double *beta,
                       double *sigsq);
void bo_ssvs_get_perm(const bo_ssvs *s, int *perm);
bo_rng *bo_ssvs_rng(bo_ssvs *s);
/* BregVsSampler::log_model_prob, BregVsSampler.cpp:216-239 */
double bo_ssvs_log_model_prob(bo_ssvs *s, const uint8_t *gamma, int *status);
/* BregVsSampler::draw, BregVsSampler.cpp:252-261.  Returns BO_OK or an error. */
int bo_ssvs_draw(bo_ssvs *s);
/* BregVsSampler::logpri, BregVsSampler.cpp:380-393, at the current state */
double bo_ssvs_logpri(bo_ssvs *s);
/* smallest |log(u) - (logp_new - logp_old)| seen over all flips so far */
double bo_ssvs_min_margin(const bo_ssvs *s);
/* replace the sufficient statistics that change under the state-space model */
void bo_ssvs_set_suf(bo_ssvs *s, const double *xty, double yty, double n,
                     double sumy, const double *xsum);

/* Convenience-ctor prior assembly (BregVsSampler.cpp:48-85 and :87-142).
 * Outputs b (p), ominv (p*p), pi (p), prior_df, sigma_guess. */
void bo_breg_prior_ctor1(int p, const double *xtx, double yty, double n,
                         double sumy, double prior_nobs, double expected_rsq,
                         double expected_model_size,
                         int first_term_is_intercept, double *b, double *ominv,
                         double *pi, double *prior_df, double *sigma_guess);
void bo_breg_prior_ctor2(int p, const double *xtx, double n, double sumy,
                         double prior_sigma_nobs, double prior_sigma_guess,
                         double prior_beta_nobs, double diagonal_shrinkage,
                         double prior_inclusion_probability,
                         int force_intercept, double *b, double *ominv,
                         double *pi, double *prior_df, double *sigma_guess);

/* Many independent chains (the engine's semantics): chain c uses the Philox
 * stream (seed, c, stream 0).  Runs nsweeps sweeps of every chain with
 * nthreads OpenMP-free pthreads (cpu_baseline leg); outputs the final state of
 * every chain.  gamma: chains x p, beta: chains x p, sigsq: chains. */
int bo_ssvs_run_chains(int p, const double *xtx, const double *xty, double yty,
                       double n, double sumy, const double *xsum,
                       const double *prior_mean, const double *ominv,
                       double prior_df, double sigma_guess, const double *pi,
                       int64_t max_model_size, double sigma_upper_limit,
                       double swap_threshold, int max_flips, uint64_t seed,
                       int chains, int nsweeps, int nthreads, uint8_t *gamma,
                       double *beta, double *sigsq);

/* ------------------------------------- SpikeSlabSampler (sigma^2 given) */
/* The sigma^2-conditional SSVS helper, Models/Glm/PosteriorSamplers/
 * SpikeSlabSampler.cpp:40-82 (draw_inclusion_indicators), :115-138
 * (draw_coefficients_given_inclusion), :171-203 (log_model_prob), :205-216.
 * slab_kind 0: precision `prec` independent of sigma^2 (MvnModel);
 * slab_kind 1: precision prec / sigsq (MvnGivenScalarSigma::siginv,
 * MvnGivenScalarSigma.cpp:74-77).  xtx / xty are (weighted) sufficient
 * statistics (WeightedRegSuf). */
typedef struct bo_sss bo_sss;
bo_sss *bo_sss_create(int p, const double *xtx, const double *xty, int slab_kind,
                      const double *mu, const double *prec, const double *pi);
void bo_sss_destroy(bo_sss *s);
void bo_sss_set_options(bo_sss *s, int64_t max_model_size, int max_flips);
void bo_sss_set_state(bo_sss *s, const uint8_t *gamma, const double *beta);
void bo_sss_get_state(const bo_sss *s, uint8_t *gamma, double *beta);
bo_rng *bo_sss_rng(bo_sss *s);
double bo_sss_log_model_prob(bo_sss *s, const uint8_t *gamma, double sigsq);
int bo_sss_draw_model_indicators(bo_sss *s, double sigsq);
int bo_sss_draw_beta(bo_sss *s, double sigsq);

/* ---------------------------------------------------------- state space */
typedef struct bo_ss bo_ss;

/* StateSpaceRegressionModel(y, X, observed) + one LocalLevelStateModel
 * (StateSpaceRegressionModel.cpp:100-125, LocalLevelStateModel.cpp:32-91).
 * X is T x p column-major; observed may be NULL (all observed). */
bo_ss *bo_ss_create(int T, int p, const double *y, const double *X,
                    const uint8_t *observed, const double *prior_mean,
                    const double *ominv, double prior_df, double sigma_guess,
                    const double *pi, double level_df,
                    double level_sigma_guess, double level_sigma_upper_limit,
                    double initial_state_mean, double initial_state_variance,
                    double initial_level_sigma);
void bo_ss_destroy(bo_ss *m);
bo_ssvs *bo_ss_regression(bo_ss *m);
bo_rng *bo_ss_level_rng(bo_ss *m);
bo_rng *bo_ss_state_rng(bo_ss *m);
void bo_ss_set_level_sigsq(bo_ss *m, double sigsq);
double bo_ss_level_sigsq(const bo_ss *m);
const double *bo_ss_state(const bo_ss *m);
void bo_ss_level_suf(const bo_ss *m, double *n, double *sumsq);
/* Base::impute_state, StateSpaceModelBase.cpp:278-291 */
int bo_ss_impute_state(bo_ss *m, bo_rng *rng);
/* StateSpacePosteriorSampler::draw, StateSpacePosteriorSampler.cpp:42-64 */
int bo_ss_draw(bo_ss *m);

/* simulate_forecast of the local level + regression model
 * (StateSpaceRegressionModel.cpp:214-219, :256-278); newX horizon x p column-major */
void bo_ss_simulate_forecast(bo_rng *rng, int horizon, int p, const double *newX,
                             const double *beta, double sigsq_obs,
                             double sigsq_level, double final_state, double *out);

/* Structural time series: regression + any list of state models
 * in any order (StateSpaceModelBase::add_state): LocalLevelStateModel,
 * LocalLinearTrendStateModel (one ZeroMeanMvnIndependenceSampler per variance),
 * SeasonalStateModel(nseasons, season_duration), ArStateModel(lags); state dimension
 * <= 64, at most 8 blocks.  bo_ssm_create is the template of rounds 2-3 (trend = 1:
 * local level; 2: local linear trend; + optional SeasonalStateModel(nseasons, 1);
 * three-element arrays indexed level, slope, seasonal); bo_ssm_create_empty +
 * bo_ssm_add_block build any list. */
enum { BO_BLK_LOCAL_LEVEL = 1, BO_BLK_LOCAL_LINEAR_TREND = 2, BO_BLK_SEASONAL = 3, BO_BLK_AR = 4,
       /* round 6 (VERDICT r5 task 8): StaticInterceptStateModel (no parameter: the var_*
        * arrays are not read), TrigStateModel (iparams = {number of frequencies}; the
        * rotations' (cos, sin) pairs, two doubles per frequency, go in through
        * initial_phi; one variance for all its components) */
       BO_BLK_STATIC_INTERCEPT = 5, BO_BLK_TRIG = 6,
       /* SemilocalLinearTrendStateModel (level, slope, the slope's long-run mean): iparams =
        * {force_stationary, force_ar1_positive}; var_* = (level, slope); initial_phi = {slope mean
        * prior mu, sigma, slope AR(1) prior mu, sigma, initial mu, initial phi}; its coefficients
        * come back through bo_ssm_block_get's phi: (phi, mu) */
       BO_BLK_SEMILOCAL = 7 };
typedef struct bo_ssm bo_ssm;
bo_ssm *bo_ssm_create(int T, int p, const double *y, const double *X,
                      const uint8_t *observed, const double *prior_mean,
                      const double *ominv, double prior_df, double sigma_guess,
                      const double *pi, int trend, int nseasons,
                      const double *var_df, const double *var_sigma_guess,
                      const double *var_sigma_upper_limit,
                      const double *var_initial_sigma,
                      const double *initial_state_mean,
                      const double *initial_state_variance);
bo_ssm *bo_ssm_create_empty(int T, int p, const double *y, const double *X,
                            const uint8_t *observed, const double *prior_mean,
                            const double *ominv, double prior_df, double sigma_guess,
                            const double *pi);
/* model->add_state(...): iparams = {nseasons, season_duration, time_of_first_observation}
 * (seasonal) or {lags} (autoregression); one entry per variance parameter in the var_*
 * arrays (two for the local linear trend: level, slope); the block's initial state is
 * N(mean, diag(variance)) */
int bo_ssm_add_block(bo_ssm *m, int kind, const int *iparams, const double *var_df,
                     const double *var_sigma_guess, const double *var_sigma_upper_limit,
                     const double *var_initial_sigma, const double *initial_phi,
                     const double *initial_state_mean, const double *initial_state_variance);
int bo_ssm_nblocks(const bo_ssm *m);
/* the generator of block b's v-th variance sampler (autoregression: the ArPosteriorSampler's) */
bo_rng *bo_ssm_block_rng(bo_ssm *m, int b, int v);
/* its Philox sampler id: level 1, slope 6, seasonal 7, autoregression 12, + 16 for every
 * earlier block of the same family */
int bo_ssm_block_stream_id(const bo_ssm *m, int b, int v);
void bo_ssm_block_get(const bo_ssm *m, int b, double *sigsq, double *suf_n, double *suf_ss,
                      double *phi);
void bo_ssm_block_set_sigsq(bo_ssm *m, int b, const double *sigsq);
void bo_ssm_block_get_ar_suf(const bo_ssm *m, int b, double *xtx, double *xty, double *yty,
                             double *n);
/* simulate_forecast from the model's current parameters */
void bo_ssm_forecast_model(const bo_ssm *m, bo_rng *rng, int horizon, int p, const double *newX,
                           const double *beta, double sigsq_obs, const double *final_state,
                           double *out);
/* adds an ArStateModel(lags) block (+ ArPosteriorSampler) after the trend / seasonal blocks */
int bo_ssm_add_ar(bo_ssm *m, int lags, double prior_df, double sigma_guess,
                  double sigma_upper_limit, double initial_sigma, const double *initial_phi,
                  const double *initial_state_mean, const double *initial_state_variance);
bo_rng *bo_ssm_ar_rng(bo_ssm *m);
/* MT mode: the restated GlobalRng::rng, which draw_phi's proposals use (rmvn_ivar) */
void bo_ssm_set_global_rng(bo_ssm *m, bo_rng *global);
void bo_ssm_get_ar(const bo_ssm *m, double *phi, double *sigsq);
void bo_ssm_get_ar_suf(const bo_ssm *m, double *xtx, double *xty, double *yty, double *n);
int bo_test_ar_check_stationary(int L, const double *phi);
/* rtrun_norm_2_mt(rng, mu, sigma, lo, hi), lo and hi finite (trun_norm.cpp:273-325, Tn2Sampler.cpp) */
double bo_rtrun_norm_2(bo_rng *rng, double mu, double sigma, double lo, double hi, int *status);
void bo_ssm_simulate_forecast_ar(bo_rng *rng, int horizon, int p, const double *newX,
                                 const double *beta, double sigsq_obs, int trend, int nseasons,
                                 const double *sigsq, int ar_lags, const double *phi,
                                 double ar_sigsq, const double *final_state, double *out);
void bo_ssm_destroy(bo_ssm *m);
bo_ssvs *bo_ssm_regression(bo_ssm *m);
bo_rng *bo_ssm_variance_rng(bo_ssm *m, int which);
bo_rng *bo_ssm_state_rng(bo_ssm *m);
int bo_ssm_state_dimension(const bo_ssm *m);
const double *bo_ssm_state(const bo_ssm *m);   /* m x T, column t = state at t */
void bo_ssm_get_variances(const bo_ssm *m, double *sigsq);
void bo_ssm_set_variances(bo_ssm *m, const double *sigsq);
void bo_ssm_get_suf(const bo_ssm *m, double *n, double *ss);
int bo_ssm_impute_state(bo_ssm *m, bo_rng *rng);
int bo_ssm_draw(bo_ssm *m);
void bo_ssm_simulate_forecast(bo_rng *rng, int horizon, int p, const double *newX,
                              const double *beta, double sigsq_obs, int trend, int nseasons,
                              const double *sigsq, const double *final_state, double *out);

/* BinomialProbitSpikeSlabSampler (probit): truncated-normal
 * data augmentation + SpikeSlabSampler on X'NX (fixed) and X'z.  X is n x p
 * column-major, y successes, ntrials trials; the slab is a fixed-precision
 * MvnModel(mu, prec). */
typedef struct bo_probit bo_probit;
bo_probit *bo_probit_create(int n, int p, const double *X, const double *y,
                            const double *ntrials, const double *mu, const double *prec,
                            const double *pi, int clt_threshold);
void bo_probit_destroy(bo_probit *m);
bo_sss *bo_probit_sss(bo_probit *m);           /* state, options, the sampler's RNG */
bo_rng *bo_probit_imputer_rng(bo_probit *m);
/* 0: the imputation reads the sampler's RNG in sequence (the reference);
 * 1: observation i of sweep s reads the imputer stream at (s n + i) * 256 */
void bo_probit_use_substreams(bo_probit *m, int on);
int bo_probit_draw(bo_probit *m);
double bo_rtrun_norm(bo_rng *r, double mu, double sigma, double a, int gt, int *status);

/* BinomialLogitSpikeSlabSampler (logit; auxiliary-mixture data
 * augmentation + its own inclusion / coefficient draws on X'WX, X'Wz) */
typedef struct bo_logit bo_logit;
bo_logit *bo_logit_create(int n, int p, const double *X, const double *y,
                          const double *ntrials, const double *mu, const double *prec,
                          const double *pi, int clt_threshold);
void bo_logit_destroy(bo_logit *m);
bo_sss *bo_logit_sss(bo_logit *m);
bo_rng *bo_logit_worker_rng(bo_logit *m);
void bo_logit_use_substreams(bo_logit *m, int on);
/* 0: the reference's auxiliary-mixture imputer (default); 1: Polya-Gamma augmentation (no
 * reference: Polson, Scott and Windle 2013; the device's CPU twin) */
void bo_logit_set_imputer(bo_logit *m, int kind);
void bo_logit_get_suf(const bo_logit *m, double *xtx, double *xty);
int bo_logit_draw(bo_logit *m);
void bo_sss_set_shuffle_kind(bo_sss *s, int kind);

/* AdaptiveSpikeSlabRegressionSampler on top of a bo_ssvs
 * (AdaptiveSpikeSlabRegressionSampler.cpp:62-225) */
typedef struct bo_adaptive bo_adaptive;
bo_adaptive *bo_adaptive_create(bo_ssvs *s);
void bo_adaptive_destroy(bo_adaptive *a);
void bo_adaptive_set_options(bo_adaptive *a, int max_flips, double step,
                             double target);
void bo_adaptive_get_rates(const bo_adaptive *a, double *birth, double *death);
double bo_adaptive_min_margin(const bo_adaptive *a);
double bo_adaptive_min_multi_margin(const bo_adaptive *a);
int bo_adaptive_draw(bo_adaptive *a);


/* ---- PoissonRegressionSpikeSlabSampler (f3, Poisson member): the mixtures of the
 * reference's NegLogGamma table for the counts in the data are an INPUT */
typedef struct bo_poisson bo_poisson;
bo_poisson *bo_poisson_create(int n, int p, const double *X, const double *y, const double *exposure,
                              const double *mu, const double *prec, const double *pi, int ncounts,
                              const int64_t *counts, const int *ncomp, const double *mix_mu,
                              const double *mix_sigma, const double *mix_weight, int64_t largest_index);
void bo_poisson_destroy(bo_poisson *m);
bo_sss *bo_poisson_sss(bo_poisson *m);
bo_rng *bo_poisson_worker_rng(bo_poisson *m);
void bo_poisson_use_substreams(bo_poisson *m, int on);
void bo_poisson_get_suf(const bo_poisson *m, double *xtx, double *xty);
int bo_poisson_draw(bo_poisson *m);

#ifdef __cplusplus
}
#endif
#endif
