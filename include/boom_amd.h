/* boom_amd -- C-ABI of the MI355X many-chain engine for BOOM's spike-and-slab
 * (BregVsSampler) and bsts local-level + regression (StateSpacePosteriorSampler)
 * hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types, no exceptions.  Every entry point names the reference interface it
 * stands in for (paths relative to the BOOM tree).  INTEGRATION.md shows the
 * binding a BOOM maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns BA_OK (0) or a negative BA_E_* code;
 *     ba_last_error() gives the message (same wording as the reference's
 *     report_error() text where one exists, cpputil/report_error.cpp:30-32);
 *   - matrices are column-major doubles exactly like BOOM::Matrix
 *     (LinAlg/Matrix.hpp:429); host pointers unless the name says _device;
 *   - caller owns every buffer it passes; the engine copies on call;
 *   - an engine lives on ONE HIP device and runs `chains` independent chains,
 *     global chain ids chain_offset .. chain_offset+chains-1 (the id keys the
 *     Philox stream, so a job sharded over ranks draws the same numbers as the
 *     same job on one device);
 *   - "chain = -1" addresses all chains at once.
 */
#ifndef BOOM_AMD_H
#define BOOM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  BA_OK = 0,
  BA_E_INVALID = -1,       /* bad argument / dimension mismatch */
  BA_E_HIP = -2,           /* HIP runtime failure (message has the HIP error) */
  BA_E_NOT_PD = -3,        /* "The posterior information matrix is not positive
                              definite..." BregVsSampler.cpp:339-350 */
  BA_E_NEGATIVE_SS = -4,   /* "Illegal data caused negative sum of squares..."
                              BregVsSampler.cpp:448-478 */
  BA_E_ILLEGAL_START = -5, /* "BregVsSampler did not start with a legal
                              configuration." BregVsSampler.cpp:364-370 */
  BA_E_RNG_BRANCH = -6,    /* a sigma^2 draw the reference itself reports as an
                              error: truncation point exactly at the gamma's mode
                              (BoundedAdaptiveRejectionSampler.cpp:50-59), or the
                              samplers' attempt limits (1000 rejections; 64 hull
                              points here) */
  BA_E_FORECAST_VARIANCE = -7, /* "Found a zero (or negative) forecast
                              variance!" ScalarKalmanFilter.cpp:48-52 */
  BA_E_MODEL_TOO_LARGE = -8,   /* model size exceeded the engine's capacity */
  BA_E_STATE = -9          /* call sequence error (e.g. sweep before priors) */
};

typedef struct ba_engine ba_engine;

typedef struct ba_config {
  int32_t device;        /* HIP device ordinal */
  int32_t chains;        /* chains resident on this device */
  int64_t chain_offset;  /* global id of local chain 0 */
  uint64_t seed;         /* sampler seed: PosteriorSampler::set_seed,
                            Models/PosteriorSamplers/PosteriorSampler.hpp:60 */
  int32_t max_model_size_hint; /* <=0: engine chooses its working capacity */
  int32_t reserved;
} ba_config;

/* message of the last failing call on this thread (never NULL) */
const char *ba_last_error(void);

/* ---- engine lifetime ---------------------------------------------------- */
int ba_engine_create(const ba_config *cfg, ba_engine **out);
void ba_engine_destroy(ba_engine *e);
/* device ordinal, chain count, predictors (0 until data are set) */
int ba_engine_info(const ba_engine *e, int32_t *device, int32_t *chains,
                   int32_t *p);

/* ---- data: RegressionModel / NeRegSuf ------------------------------------ */
/* NeRegSuf(X, y), Models/Glm/RegressionModel.cpp:309-328: builds XtX, Xty,
 * yty, n, sum(y), column sums of X on the device (f64 MFMA syrk).  X is n x p
 * column-major. */
int ba_build_suf_from_xy(ba_engine *e, int64_t n, int32_t p, const double *X,
                         const double *y);
/* same, X / y already resident in device memory of e's device */
int ba_build_suf_from_xy_device(ba_engine *e, int64_t n, int32_t p,
                                const void *X_device, const void *y_device);
/* The same build for a design matrix SHARDED BY ROWS over the ranks of a job
 * (config 4: X = 3.3 GB): every rank computes the statistics of its own rows
 * into a device block of ba_suf_block_size(p) doubles laid out as
 *   [ X'X (p x p, column-major) | X'y (p) | y'y, sum y | column sums of X (p) ],
 * the caller sums the blocks over ranks -- ONE all-reduce (RCCL over xGMI;
 * boom_amd/dist.py: build_suf_row_sharded) -- and every rank installs the total.
 * The sums are what NeRegSuf(X, y) holds (RegressionModel.cpp:309-328); every
 * rank ends up with bitwise the same statistics, so chains do not depend on
 * which rank runs them.  X_device is the shard, n_rows x p column-major. */
size_t ba_suf_block_size(int32_t p);
int ba_suf_partial_device(ba_engine *e, int64_t n_rows, int32_t p,
                          const void *X_device, const void *y_device,
                          void *block_device);
int ba_set_suf_from_block_device(ba_engine *e, int64_t n_total, int32_t p,
                                 const void *block_device);
/* NeRegSuf(XTX, XTY, YTY, n, ybar, xbar), RegressionModel.cpp:330-345 */
int ba_upload_regression_suf(ba_engine *e, int32_t p, const double *xtx,
                             const double *xty, double yty, double n,
                             double ybar, const double *xbar);
/* RegSuf accessors xtx()/xty()/yty()/n()/ybar()/xbar(),
 * RegressionModel.hpp:58-101.  Any output pointer may be NULL. */
int ba_get_regression_suf(ba_engine *e, double *xtx, double *xty, double *yty,
                          double *n, double *ybar, double *xbar);

/* ---- priors: BregVsSampler ctor #5 pieces -------------------------------- */
/* slab: MvnGivenScalarSigma(b, Omega^{-1}) (Models/MvnGivenScalarSigma.hpp:57-109;
 * BregVsSampler::set_slab, BregVsSampler.hpp:111-113) */
int ba_set_slab(ba_engine *e, const double *prior_mean,
                const double *unscaled_prior_precision);
/* spike: VariableSelectionPrior(pi) + set_max_model_size (< 0: none)
 * (VariableSelectionPrior.hpp:101,133; BregVsSampler::set_spike) */
int ba_set_spike(ba_engine *e, const double *prior_inclusion_probabilities,
                 int64_t max_model_size);
/* residual precision prior ChisqModel(df, sigma_guess) + set_sigma_upper_limit
 * (ChisqModel.cpp:56-57; BregVsSampler.cpp:264-266); sigma_upper_limit = +inf
 * for none */
int ba_set_sigma_prior(ba_engine *e, double prior_df, double sigma_guess,
                       double sigma_upper_limit);
/* convenience ctors #1 and #2 (BregVsSampler.cpp:48-85, :87-142): assemble the
 * three priors above from the current sufficient statistics */
int ba_set_priors_ctor1(ba_engine *e, double prior_nobs, double expected_rsq,
                        double expected_model_size,
                        int32_t first_term_is_intercept);
int ba_set_priors_ctor2(ba_engine *e, double prior_sigma_nobs,
                        double prior_sigma_guess, double prior_beta_nobs,
                        double diagonal_shrinkage,
                        double prior_inclusion_probability,
                        int32_t force_intercept);
/* read back the assembled priors (any pointer may be NULL) */
int ba_get_priors(ba_engine *e, double *prior_mean, double *ominv, double *pi,
                  double *prior_df, double *prior_ss);

/* limit_model_selection / suppress_model_selection (max_flips: <0 = p, 0 =
 * none), set_correlation_swap_threshold, suppress_beta_draw /
 * suppress_sigma_draw (BregVsSampler.hpp:131-161) */
int ba_set_options(ba_engine *e, int32_t max_flips, double swap_threshold,
                   int32_t draw_beta, int32_t draw_sigma);

/* Overrides of choices the engine otherwise makes itself; never needed for
 * correctness (every setting produces the same chains), used by the tests to
 * force every code path and by diagnostics:
 *   waves_per_chain  0 = engine chooses; 1, 2 or 4 wavefronts per chain
 *   walk_policy      -1 = default (adaptive); 0 batch mode only, 1 adaptive,
 *                    2 always the proposal table, 3 adaptive without forked
 *                    quiet sweeps
 *   kcap_start       0 = default; first model capacity tried (16, 32, ...) */
int ba_set_tuning(ba_engine *e, int32_t waves_per_chain, int32_t walk_policy,
                  int32_t kcap_start);
/* The samplers that give every draw its own slot of a stream (the bsts state draw's normals:
 * 256 positions a normal; the probit / logit / Polya-Gamma / Poisson imputers: 4096 / 256 /
 * 4096 / 256 positions an observation) let a draw that needs more uniforms than its slot
 * holds go on in the slot's SPILL stream (same chain, stream id | 0x80000000, position
 * slot << 20) -- an event of probability < 1e-40 at these strides.  For the tests of that
 * path: a slot hands out only `uniforms` numbers (even; 0 = the whole stride again), so
 * that the spill streams are read all the time.  The oracle's bo_set_slot_limit is the
 * same switch.  Changes the draws. */
int ba_set_slot_limit(ba_engine *e, int32_t uniforms);

/* ---- chain state: GlmCoefs (beta, inc) + sigsq ----------------------------- */
/* coef().set_inc / set_Beta / set_sigsq.  gamma: p bytes (0/1), beta: p
 * doubles (may be NULL = zeros).  chain = -1 sets every chain. */
int ba_set_state(ba_engine *e, int64_t chain, const uint8_t *gamma,
                 const double *beta, double sigsq);
/* coef().inc() / Beta() / sigsq() of one chain (local index).  While
 * ba_draw_next() is serving a look-ahead batch these are the draw being served,
 * not the end of the batch. */
int ba_get_state(ba_engine *e, int64_t chain, uint8_t *gamma, double *beta,
                 double *sigsq);
/* all chains at once: gamma chains x p, beta chains x p, sigsq chains */
int ba_get_states(ba_engine *e, uint8_t *gamma, double *beta, double *sigsq);
/* PosteriorSampler::logpri() of BregVsSampler (BregVsSampler.cpp:380-393) at the
 * current state of one chain (local index): log p(gamma) + log p(sigma^2) +
 * log N(beta_gamma | b_gamma, sigma^2 Omega_gamma) */
int ba_logpri(ba_engine *e, int64_t chain, double *out);
/* PosteriorSampler::set_seed: re-keys the streams of every sampler of every
 * chain (regression, SpikeSlab, level, state), all at position 0 */
int ba_seed(ba_engine *e, uint64_t seed);

/* ---- the hot path ---------------------------------------------------------- */
/* nsweeps x BregVsSampler::draw() (BregVsSampler.cpp:252-261) on every chain.
 * Asynchronous: returns after the launch.  Errors raised by chains surface at
 * the next ba_sync()/ba_get_*(). */
int ba_sweep(ba_engine *e, int32_t nsweeps);
/* wait for outstanding work and report the first chain error, if any.  (The chains'
 * status words come back in one batched copy; a ba_sync() that follows a clean one with
 * nothing in between but accessors -- ba_get_state, ba_ss_get_state, ba_logpri ... , each
 * of which begins with one -- is only the wait.)  While ba_ss_draw_next is serving draws
 * from a look-ahead batch, ba_sync waits for THAT batch only: the batch running ahead of it
 * stays in flight (a chain that stops there is reported when its draws are served), and it is
 * no barrier for work the caller queued on ba_stream() -- synchronise that stream itself. */
int ba_sync(ba_engine *e);
/* The reference's calling pattern is ONE draw() per MCMC iteration with the
 * caller reading the parameters in between (spike_slab_wrapper.cc:233-242,
 * spikeslab.py:191-207; Model::sample_posterior, Models/ModelTypes.hpp:93).
 * ba_draw_next() keeps that pattern at the long-launch rate: every `lookahead`
 * calls it launches `lookahead` sweeps of EVERY chain with the draws recorded
 * on the device, the calls in between only move a cursor, and
 * ba_get_state / ba_get_states / ba_logpri return the draw under the cursor --
 * bitwise what `ba_sweep(e, 1)` per iteration leaves there.  Any other call
 * that touches the engine (data, priors, options, state, seed, ba_sweep) first
 * puts the chains back where the caller has seen them (the batch's start state
 * replayed up to the cursor), so the sequence of draws never depends on the
 * look-ahead length.  Running summaries count launched sweeps, i.e. they run
 * ahead of the cursor by up to lookahead - 1 draws.
 * ba_set_lookahead(e, n): n >= 1 (1 = one launch per call; replaces any
 * ba_enable_traces / ba_enable_draws setting). */
int ba_set_lookahead(ba_engine *e, int32_t lookahead);
int ba_draw_next(ba_engine *e);
/* log_model_prob(gamma) for ngamma inclusion vectors (BregVsSampler.cpp:216-239)
 * on the regression model's sufficient statistics, models of any size (up to 1024
 * included variables); BA_E_STATE once state-space data are set (there they are per
 * chain and move every sweep) */
int ba_log_model_prob(ba_engine *e, int32_t ngamma, const uint8_t *gammas,
                      double *out);

/* ---- SpikeSlabSampler: the sigma^2-conditional sweep ---------------------------- */
/* Models/Glm/PosteriorSamplers/SpikeSlabSampler.{hpp,cpp}: the helper that the
 * logit / probit / Poisson / Student / quantile samplers drive with their own
 * (weighted) sufficient statistics and the current residual variance.
 *   slab      SpikeSlabSampler(model, slab_prior, spike_prior) (.hpp:47): mean mu
 *             and precision; precision_scales_with_sigsq = 1 for
 *             MvnGivenScalarSigma (siginv() = Omega^{-1} / sigma^2,
 *             MvnGivenScalarSigma.cpp:74-77), 0 for a fixed-precision MvnBase
 *             (then every chain must hold the same sigma^2);
 *   max_flips limit_model_selection(max_flips) (.hpp:132): limits only if > 0;
 *   spike     ba_set_spike; data: ba_upload_regression_suf with the weighted
 *             X'WX, X'Wy (WeightedRegSuf; yty, n, ybar, xbar are not used);
 *   sigma^2   per chain, ba_set_sigsq / ba_set_state.
 * ba_sss_sweep runs nsweeps x { draw_model_indicators(rng, suf, sigsq);
 * draw_beta(rng, suf, sigsq) } (SpikeSlabSampler.cpp:40-138) on every chain. */
int ba_set_sigsq(ba_engine *e, int64_t chain, double sigsq);
int ba_sss_set_slab(ba_engine *e, const double *mu, const double *precision,
                    int32_t precision_scales_with_sigsq, int32_t max_flips);
int ba_sss_sweep(ba_engine *e, int32_t nsweeps);

/* ---- AdaptiveSpikeSlabRegressionSampler: what lm.spike runs for p > 100 -------- */
/* Models/Glm/PosteriorSamplers/AdaptiveSpikeSlabRegressionSampler.{hpp,cpp}
 * (Interfaces/R/BoomSpikeSlab/src/spike_slab_wrapper.cc:99-140): same model,
 * priors (ba_set_slab / ba_set_spike / ba_set_sigma_prior) and chain state as
 * BregVsSampler; a sweep is min(max_flips, p) birth / death moves with adaptive
 * proposal rates, then sigma^2 and beta.
 *   ba_adaptive_set_options  limit_model_selection(max_flips) (default 100; 0 =
 *                            allow_model_selection(false)), set_step_size
 *                            (.001), set_target_acceptance_rate (.345); a
 *                            negative value keeps the current setting
 *   ba_adaptive_sweep        nsweeps x draw() (.cpp:62-85) on every chain
 *   ba_adaptive_get_rates    birth_rates_, death_rates_, iteration_count_ of a
 *                            chain (any pointer may be NULL)
 * A chain whose model grows beyond 64 variables moves to the large-model kernel, as
 * BregVsSampler's chains do (its moves are then read off that kernel's table of
 * log model probabilities, rebuilt after every accepted move); the limit is the
 * large-model kernel's (1024 variables), beyond it BA_E_MODEL_TOO_LARGE. */
int ba_adaptive_set_options(ba_engine *e, int32_t max_flips, double step_size,
                            double target_acceptance_rate);
int ba_adaptive_sweep(ba_engine *e, int32_t nsweeps);
int ba_adaptive_get_rates(ba_engine *e, int64_t chain, double *birth_rates,
                          double *death_rates, uint64_t *iteration_count);

/* ---- BinomialProbitSpikeSlabSampler (the probit member) -----
 * Data augmentation (BinomialProbitDataImputer.cpp:30-73: a truncated normal per
 * trial, or the central-limit draw beyond clt_threshold trials of a kind) followed
 * by SpikeSlabSampler on the complete-data sufficient statistics X'NX (fixed) and
 * X'z (BinomialProbitSpikeSlabSampler.cpp:40-85).  X is n x p column-major, y
 * successes, ntrials trials.  Priors: ba_sss_set_slab(mu, precision,
 * scales_with_sigsq = 0, max_flips) and ba_set_spike.  State through
 * ba_set_state / ba_get_state(s) (sigma^2 is 1).  RNG: stream 3 for the
 * inclusion / coefficient draws; the imputation of observation i in sweep s reads
 * stream 8 from position (s n + i) * 4096 -- the reference reads ONE stream in
 * sequence, a counter-based one lets the observations go in parallel. */
int ba_probit_set_data(ba_engine *e, int64_t n, int32_t p, const double *X,
                       const double *y, const double *ntrials, int32_t clt_threshold);
int ba_probit_sweep(ba_engine *e, int32_t nsweeps);

/* ---- BinomialLogitSpikeSlabSampler (the logit member; the
 * sampler behind BASELINE config 5, with the reference's own auxiliary-mixture
 * imputer -- it has no Polya-Gamma one) -------------------------------------------
 * Per sweep: every trial's truncated logistic draw and mixture component
 * (BinomialLogitAuxmixSampler.cpp:77-97, BinomialLogitDataImputer.cpp:128-144),
 * X'Wz for all chains by one MFMA GEMM, of every chain's own slab precision + X'WX
 * the vectors the sweep reads (those of included variables, built by one gathered
 * MFMA GEMM; a variable that enters mid-sweep has its vector built on request and
 * the chain replays that sweep), then the sampler's inclusion / coefficient draws
 * (BinomialLogitSpikeSlabSampler.cpp:50-117, :178-226; its shuffle of the visiting
 * order differs from SpikeSlabSampler's).  Same prior setters and state accessors as
 * the probit sampler.  Observations with more than clt_threshold (1 .. 64) trials take
 * the reference's large-sample imputation (BinomialLogitCltDataImputer::
 * impute_large_sample, BinomialLogitDataImputer.cpp:155-211: multinomial counts per
 * mixture component, one normal draw); models of any size.  RNG: stream 3 for the
 * sampler, stream 9 from position (s n + i) * 256 for the imputation of observation i
 * in sweep s (two uniforms per trial, or the large-sample branch's few dozen). */
int ba_logit_set_data(ba_engine *e, int64_t n, int32_t p, const double *X,
                      const double *y, const double *ntrials, int32_t clt_threshold);
int ba_logit_sweep(ba_engine *e, int32_t nsweeps);
/* The imputation step of the logit sampler: 0 (default) the reference's auxiliary
 * mixture; 1 Polya-Gamma augmentation -- omega_i ~ PG(n_i, x_i'beta) by Devroye's exact
 * sampler (Polson, Scott and Windle 2013), the normal with PG's moments beyond
 * clt_threshold trials; (y_i - n_i / 2, omega_i) takes the place of the mixture's
 * (information-weighted sum, information), everything else is the same sweep.  BOOM has
 * NO Polya-Gamma sampler (BASELINE config 5 names one): there is nothing to be
 * bit-compared with; the chain it defines has the same stationary distribution as the
 * logit likelihood's exact posterior, which the auxiliary-mixture sampler approximates,
 * and is checked against that sampler distributionally.  RNG: stream 10, observation i
 * of sweep s from position (s n + i) * 4096. */
int ba_logit_set_imputer(ba_engine *e, int32_t kind);

/* ---- PoissonRegressionSpikeSlabSampler (the Poisson member) -----------
 * Models/Glm/PosteriorSamplers/PoissonRegressionSpikeSlabSampler.cpp:55-59: per sweep the
 * auxiliary-mixture imputation of Fruehwirth-Schnatter et al. (PoissonDataImputer.cpp:36-96:
 * the last event time inside the exposure interval ~ exposure x Beta(y, 1), the first
 * event after it, each negative log time unmixed against a normal mixture of
 * NegLogGamma(count)), then SpikeSlabSampler's inclusion / coefficient draws on the
 * complete-data sufficient statistics -- on the device the logit path's machinery: X'Wz by
 * one MFMA GEMM, every chain's own slab precision + X'WX a vector at a time.
 *   ba_poisson_set_data      X n x p column-major, y counts, exposure (> 0)
 *   ba_poisson_set_mixtures  the mixtures, BY THE CALLER: the reference keeps them in a
 *                            table that interpolates and refits on demand
 *                            (create_poisson_mixture_approximation_table /
 *                            NormalMixtureApproximationTable::approximate); the BOOM-side
 *                            binding reads its table, the tests a fixture generated from
 *                            the compiled reference.  counts ascending (must cover 1 and
 *                            every positive count in the data below largest_index, from
 *                            where on the Gaussian limit is used); mixture i has ncomp[i]
 *                            (<= 32) components, packed in mu / sigma / weight
 *   priors, state            ba_sss_set_slab(mu, precision, 0, max_flips), ba_set_spike,
 *                            ba_set_state / ba_get_state(s) as for the logit sampler
 * RNG: stream 3 for the sampler, stream 11 from position (s n + i) * 256 for the
 * imputation of observation i in sweep s. */
int ba_poisson_set_data(ba_engine *e, int64_t n, int32_t p, const double *X, const double *y,
                        const double *exposure);
int ba_poisson_set_mixtures(ba_engine *e, int32_t ncounts, const int64_t *counts,
                            const int32_t *ncomp, const double *mu, const double *sigma,
                            const double *weight, int64_t largest_index);
int ba_poisson_sweep(ba_engine *e, int32_t nsweeps);

/* ---- posterior summaries --------------------------------------------------- */
/* Running sums over every sweep since the last ba_reset_summaries(), reduced
 * over this engine's chains on the device:
 *   inclusion_count[p] (as double), beta_sum[p], beta_sumsq[p],
 *   scalars[16] = {sweeps*chains, sum sigsq, sum sigsq^2, sum |gamma|,
 *                  accepted flips, proposed flips, min accept margin, accepted
 *                  flips served from a chain's other slot,
 *                  [8..15] per-phase cycle counters of the diagnostic build
 *                  (zero in the production library; after ba_adaptive_sweep
 *                  scalar 8 is the smallest relative distance of a weighted-
 *                  draw uniform from a boundary of the cumulative rates)}
 * The layout is one contiguous block of (3p + 16) doubles so that a single
 * collective moves it (see ba_summaries_device). */
int ba_reset_summaries(ba_engine *e);
int ba_get_summaries(ba_engine *e, double *inclusion_count, double *beta_sum,
                     double *beta_sumsq, double *scalars);
/* reduce into a device buffer of (3p + 16) doubles owned by the caller (e.g. a
 * torch tensor that then goes through RCCL); stream-ordered on the engine's
 * stream followed by a sync */
int ba_summaries_device(ba_engine *e, void *out_device);
/* per-sweep traces of the last ba_sweep call for ESS: each chains x nsweeps,
 * row-major; any pointer may be NULL.  Tracing must be enabled first. */
int ba_enable_traces(ba_engine *e, int32_t max_sweeps);
int ba_get_traces(ba_engine *e, int32_t nsweeps, double *sigsq, double *logp,
                  double *model_size);
/* Recording of every sweep's draw (the step the callers do on the host today:
 * RListIoManager::write, Interfaces/R/list_io.cpp; spikeslab.py:191-207): with
 * recording enabled a ba_sweep(n) call keeps the n draws of every chain on the
 * device (traces included), so a `for (i < niter) sample_posterior(); record()`
 * loop becomes one launch and niter reads.  ba_get_draws expands one chain's
 * (local index, as everywhere) draws: gamma nsweeps x p, beta nsweeps x p
 * (zeros outside gamma), sigsq nsweeps; any pointer may be NULL. */
int ba_enable_draws(ba_engine *e, int32_t max_sweeps);
int ba_get_draws(ba_engine *e, int64_t chain, int32_t nsweeps, uint8_t *gamma,
                 double *beta, double *sigsq);

/* Posterior predictive means from the record (lm_spike.predict, spikeslab.py:530-546:
 * coefficient_draws[burn:, :] @ predictors.T), all chains at once: draws
 * [first_draw, first_draw + ndraws) of the last recorded ba_sweep call (first_draw =
 * the caller's burn-in), newX column-major nnew x p, out chains x ndraws x nnew. */
int ba_predict(ba_engine *e, int32_t first_draw, int32_t ndraws, int32_t nnew,
               const double *newX, double *out);

/* the recorded coefficient paths of chosen variables (ESS of the largest
 * coefficients): out is chains x nvars x nsweeps, 0 where a variable was
 * excluded */
int ba_get_coefficient_traces(ba_engine *e, int32_t nsweeps, int32_t nvars,
                              const int32_t *vars, double *out);

/* the engine's HIP stream (hipStream_t) for callers that order their own work: what is put
 * on it after this call runs after every launch the engine has issued (consecutive
 * ba_sweep calls overlap on a second stream of the engine; this call joins them) */
void *ba_stream(ba_engine *e);

/* ---- measurement -------------------------------------------------------------
 * Device time per kernel class (new; no reference counterpart -- the reference has
 * no profiler hook).  With timing enabled every kernel launch of the
 * engine is bracketed by a pair of HIP events on the engine's stream;
 * ba_get_kernel_times waits for the stream and returns, per class, the summed
 * milliseconds and the number of launches since the last reset (arrays of
 * ba_kernel_classes() entries; either may be NULL).  A sweep round that is several
 * kernels (bsts: SSVS + Kalman + X'e GEMM) is timed kernel by kernel this way;
 * bench.py's roofline objects come from here.  Disabled (the default) it costs
 * nothing; it never changes a draw.  enabled = 1 also keeps consecutive ba_sweep launches
 * apart (one launch at a time: clean per-kernel figures); enabled = 2 lets them overlap as
 * they do untimed (each launch's pair sits on the stream the launch went to). */
int32_t ba_kernel_classes(void);
const char *ba_kernel_class_name(int32_t cls);
int ba_set_kernel_timing(ba_engine *e, int32_t enabled);
int ba_get_kernel_times(ba_engine *e, double *ms, int64_t *launches, int32_t reset);

/* ---- bsts: StateSpaceRegressionModel + LocalLevelStateModel ----------------- */
/* StateSpaceRegressionModel(y, X, observed) (StateSpaceRegressionModel.cpp:100-125):
 * X is T x p column-major; observed may be NULL.  Replaces the engine's data:
 * XtX is fixed, per-chain Xty / yty / n are rebuilt by every impute_state. */
int ba_ss_set_data(ba_engine *e, int32_t T, int32_t p, const double *y,
                   const double *X, const uint8_t *observed);
/* LocalLevelStateModel + ZeroMeanGaussianConjSampler(df, sigma_guess) with
 * set_sigma_upper_limit, set_initial_state_mean / _variance
 * (LocalLevelStateModel.cpp:32-91; ZeroMeanGaussianConjSampler.cpp:37-60) */
int ba_ss_set_local_level(ba_engine *e, double level_df,
                          double level_sigma_guess,
                          double level_sigma_upper_limit,
                          double initial_state_mean,
                          double initial_state_variance,
                          double initial_level_sigma);
/* Richer state, instead of ba_ss_set_local_level: BOOM's
 * block-diagonal state -- any list of state models, in the order add_state receives
 * them (Models/StateSpace/StateSpaceModelBase.hpp:637-638; the transition / variance
 * matrices are block diagonal, Filters/SparseMatrix.hpp:2196).
 *
 * ba_ss_add_state_model appends ONE state model (the first call after
 * ba_ss_set_local_level / ba_ss_clear_state_models starts a new list):
 *   kind 1  LocalLevelStateModel + ZeroMeanGaussianConjSampler
 *           (StateModels/LocalLevelStateModel.cpp)                      1 component
 *   kind 2  LocalLinearTrendStateModel with one ZeroMeanMvnIndependenceSampler per
 *           variance, as bsts builds it (StateModels/LocalLinearTrend.cpp,
 *           PosteriorSamplers/ZeroMeanMvnIndependenceSampler.cpp:63-70)  2 components
 *   kind 3  SeasonalStateModel(nseasons = iparams[0], season_duration = iparams[1])
 *           with set_time_of_first_observation(iparams[2]) and a
 *           ZeroMeanGaussianConjSampler (StateModels/SeasonalStateModel.cpp: the
 *           transition / state-error variance are the seasonal matrices on the steps
 *           INTO a new season, :89-104, :248-258, and identity / zero inside one)
 *                                                                        nseasons - 1
 *   kind 4  ArStateModel(lags = iparams[0] <= 16) + ArPosteriorSampler(ChisqModel(df,
 *           sigma_guess)) (StateModels/ArStateModel.cpp; Models/TimeSeries/
 *           PosteriorSamplers/ArPosteriorSampler.cpp:52-143: up to three multivariate
 *           proposals for phi, accepted when stationary, else one coefficient at a time
 *           from a truncated normal -- the reference's Tn2Sampler on the device,
 *           distributions/Tn2Sampler.cpp:25-131 --; then sigma)          lags
 *   kind 5  StaticInterceptStateModel (StateModels/StaticInterceptStateModel.hpp:35-131,
 *           .cpp:29-54): T = 1, no state error, no parameter and no sampler -- the var_*
 *           arrays are not read (may be NULL); its value is its initial draw
 *           N(initial_state_mean, initial_state_variance >= 0), moved by the smoother  1
 *   kind 6  TrigStateModel(period, frequencies) (StateModels/TrigStateModel.cpp:130-223):
 *           iparams[0] = the number of frequencies (<= 32); initial_phi = the 2 x 2 rotations
 *           of the transition matrix, (cos, sin) of 2 pi f / period per frequency, as
 *           state_transition_matrix(0) holds them (the caller computes them -- or reads them
 *           off the reference's model object -- so that no second cosine routine is involved);
 *           Z = 1 at every pair's first component; ONE variance for all components with its
 *           ZeroMeanGaussianConjSampler(ChisqModel(df, sigma_guess)), as bsts builds it
 *           (Interfaces/R/bsts/src/create_state_model.cpp:559-586)          2 x frequencies
 *   kind 7  SemilocalLinearTrendStateModel(level, slope) (StateModels/SemilocalLinearTrend.cpp:
 *           37-323): state (level, slope, the slope's long-run mean mu); level' = level + slope +
 *           e1, slope' = mu + phi (slope - mu) + e2.  The level's variance has a
 *           ZeroMeanGaussianConjSampler, the slope's NonzeroMeanAr1Model (mu, phi, sigma) a
 *           NonzeroMeanAr1Sampler (Models/TimeSeries/PosteriorSamplers/NonzeroMeanAr1Sampler.cpp:
 *           58-137: mu | phi, sigma normal; phi | mu, sigma normal, truncated to (-1, 1) -- or
 *           (0, 1) -- by rejection when forced stationary (positive); sigma from the residual sum
 *           of squares), as bsts builds it (create_state_model.cpp:600-672).  iparams =
 *           {force_stationary, force_positive} (positive without stationary -- a one-sided
 *           truncation -- is refused); the var_* arrays have two entries (level, slope);
 *           initial_phi = {mu's prior mean, sd, phi's prior mean, sd, initial mu, initial phi};
 *           initial_state_mean / _variance have three entries, the third is not read (the
 *           component IS mu: mean mu, variance 0).  Takes one of the 4 autoregression slots.  3
 * iparams is ignored for kinds 1, 2 and 5 (may be NULL).  The var_* arrays hold one entry
 * per variance parameter of the model (two for kinds 2 and 7: level, slope; one otherwise):
 * ChisqModel(df, sigma_guess) prior, sigma upper limit (infinity: none), initial sigma.
 * initial_phi: lags entries, NULL = zeros; must be stationary, as ArModel's constructor
 * demands (ArModel::check_stationary, ArModel.cpp:142-170, decided by the quick bound
 * sum |phi| < 1 and then, where the reference finds polynomial roots, by the equivalent
 * step-down recursion).  initial_state_mean / _variance: the model's components (the
 * variance's diagonal; positive, a local level's may be 0).  Limits: state dimension
 * <= 64, 8 state models, 16 variance parameters, 4 autoregression models.
 * RNG streams: variance parameter v of a model reads the chain's sampler id 1 (level),
 * 6 (slope), 7 (seasonal), 13 (trig), 14 (a semilocal slope: NonzeroMeanAr1Sampler) or 12 (ArPosteriorSampler: the proposals' normals, then the
 * sigma draw -- the reference takes the proposals from GlobalRng::rng and the rest from
 * the sampler's generator) + 16 for every earlier model of the same family (local level
 * and local linear trend are one family); the state draw reads stream 2. */
int ba_ss_clear_state_models(ba_engine *e);
int ba_ss_add_state_model(ba_engine *e, int32_t kind, const int32_t *iparams,
                          const double *var_df, const double *var_sigma_guess,
                          const double *var_sigma_upper_limit,
                          const double *var_initial_sigma, const double *initial_phi,
                          const double *initial_state_mean,
                          const double *initial_state_variance);
/* which kernel draws the state (changes no draw): 0 = the general kernel, one chain per
 * workgroup; (2, four chains per wavefront, was removed in round 5: bit-identical and never
 * faster -- BA_E_INVALID now); 3 = the kernel compiled for the shape [level | trend] [+ seasonal of duration 1]
 * [+ one autoregression] with m <= 16 where the list has that shape; 1 = the default choice.
 * Independently of those: 4 = the local-level model's rounds as separate launches per round
 * (regression sweep, state draw, X'e), 5 = as one persistent launch per call in which every
 * chain loops over the rounds by itself (the default where it applies: a series of at most
 * 2048 steps, models of at most 48 variables).  And: 6 / 7 = the persistent launch's notes of a
 * debugging session (printed to stderr when one of this engine's chains stops) on / off (off
 * by default; the buffer is the engine's own, on its device) */
int ba_ss_set_tuning(ba_engine *e, int32_t kernel);
/* the state dimension and the number of state models of the specification */
int ba_ss_state_dimension(ba_engine *e, int32_t *state_dimension, int32_t *nblocks);
/* state model `block` of one chain: its variance parameters (nvar), the model's
 * sufficient statistics of the last sweep (n, sum of squares per variance) and, for an
 * autoregression model, its coefficients (lags) and the ArModel's sufficient
 * statistics (xtx lags x lags column-major, xty, yty, n).  A semilocal linear trend (kind 7):
 * variances = (level, slope), phi[0..1] = (phi, mu) of the slope's NonzeroMeanAr1Model, ar_xtx
 * [0..5] = its Ar1Suf (sum of squares, sum, cross product of successive values, n, first value,
 * last value; Models/TimeSeries/NonzeroMeanAr1Model.cpp:33-91), ar_n = n; ar_xty / ar_yty are
 * refused.  Any pointer may be NULL */
int ba_ss_get_state_model(ba_engine *e, int64_t chain, int32_t block, double *variances,
                          double *suf_n, double *suf_ss, double *phi, double *ar_xtx,
                          double *ar_xty, double *ar_yty, double *ar_n);
/* one chain's state draw (T x m, step t at [t * m, (t + 1) * m), the models' components
 * in the order the models were added) */
int ba_ss_get_state_draw(ba_engine *e, int64_t chain, double *state);
/* The template of rounds 2-3, kept: a trend state model -- trend = 1: local level; 2:
 * local linear trend -- plus an optional SeasonalStateModel(nseasons, season_duration =
 * 1); nseasons = 0: none.  Equivalent to ba_ss_add_state_model(trend) [+ (seasonal)].
 * The three-element arrays are indexed level, slope, seasonal (prior df, sigma guess,
 * sigma upper limit, initial sigma of each variance parameter; unused entries are
 * ignored); initial_state_mean / _variance have m = trend + max(nseasons - 1, 0)
 * entries.  RNG streams: 1 level, 6 slope, 7 seasonal, 2 state. */
int ba_ss_set_structural(ba_engine *e, int32_t trend, int32_t nseasons,
                         const double *var_df, const double *var_sigma_guess,
                         const double *var_sigma_upper_limit,
                         const double *var_initial_sigma,
                         const double *initial_state_mean,
                         const double *initial_state_variance);
/* After ba_ss_set_structural: appends ONE ArStateModel(lags) (kind 4 above) to the
 * template; ba_ss_get_ar reads it.  RNG stream 12. */
int ba_ss_add_ar(ba_engine *e, int32_t lags, double prior_df, double sigma_guess,
                 double sigma_upper_limit, double initial_sigma,
                 const double *initial_phi, const double *initial_state_mean,
                 const double *initial_state_variance);
/* one chain's autoregression coefficients (lags), error variance and the ArModel's
 * sufficient statistics of the last sweep (xtx lags x lags column-major, xty, yty,
 * n); any pointer may be NULL */
int ba_ss_get_ar(ba_engine *e, int64_t chain, double *phi, double *sigsq,
                 double *suf_xtx, double *suf_xty, double *suf_yty, double *suf_n);
/* (the template's accessor) one chain's state draw (T x m, step t at [t * m, (t + 1) * m)),
 * the three variance parameters level / slope / seasonal and the state models'
 * sufficient statistics (n, sum of squares) of the last sweep; any pointer may be NULL */
int ba_ss_get_structural(ba_engine *e, int64_t chain, double *state,
                         double *variances, double *suf_n, double *suf_ss);
/* The callers' loop on the bsts path -- for (i in niter) { model->sample_posterior();
 * record the draw } (Interfaces/R/bsts/src/bsts.cc:82-119) -- at the device's rate.
 * ba_ss_set_lookahead(e, L), L > 1: ba_ss_draw_next enqueues the sweep rounds L at a
 * time (the batch after the one being served goes out as soon as serving starts), every
 * round's draw is recorded on the device and handed out one per call: ba_get_state,
 * ba_get_states, ba_logpri, ba_ss_get_state, ba_ss_get_structural, ba_ss_get_state_model,
 * ba_ss_get_state_draw and ba_ss_get_ar see the draw being served -- gamma, beta, sigma^2
 * and the state models' variances / coefficients of EVERY chain, the state path of the
 * chains named with ba_ss_lookahead_chains (default: chain 0).  Anything else -- a
 * mutator, ba_ss_sweep, a sufficient statistic, the state path of another chain, a
 * forecast -- first puts the chains back where the caller has seen them (the snapshot
 * taken at the batch's start, replayed up to the draw being served: same stream
 * positions, same draws), so the look-ahead is unobservable.  L = 1 (default): one
 * round per call.  ba_sync while a batch is being served waits for THAT batch.
 * What such a rewind costs is bounded: L is the LARGEST batch -- every rewind halves the
 * batches that follow (down to one round per call: no look-ahead), every batch served to its
 * end doubles them again; a chain whose state path was asked for joins the recorded chains
 * (up to 32).  L is cut down to what a 2 GiB record holds. */
int ba_ss_set_lookahead(ba_engine *e, int32_t lookahead);
int ba_ss_lookahead_chains(ba_engine *e, int32_t nchains, const int64_t *chains);
int ba_ss_draw_next(ba_engine *e);
/* nsweeps x StateSpacePosteriorSampler::draw()
 * (StateSpacePosteriorSampler.cpp:42-64) on every chain.  Local level model, a series of
 * at most 2048 steps, models of at most 48 variables: the rounds of the call run as ONE
 * persistent launch per 64 rounds (every chain loops over its rounds by itself; chains
 * meet in tiles of 16 for X'(y - state)); otherwise three launches per round (regression
 * sweep, state draw, X'(y - state)).  The same draws either way: ba_ss_set_tuning(e, 4 | 5). */
int ba_ss_sweep(ba_engine *e, int32_t nsweeps);
/* one Base::impute_state (StateSpaceModelBase.cpp:278-291) with the current
 * parameters, on every chain */
int ba_ss_impute_state(ba_engine *e);
/* StateSpaceRegressionModel::simulate_forecast(rng, newX, final_state)
 * (StateSpaceRegressionModel.cpp:214-219, :256-278; what bsts' predict does for
 * every saved draw): one draw from the predictive distribution of the next
 * `horizon` observations for EVERY chain's current parameters and final state.
 * newX is horizon x p column-major (host); out is chains x horizon (host),
 * row-major.  Each call continues the chains' forecast streams.  Works for the
 * local level model and for the structural (trend + seasonal) one. */
int ba_ss_forecast(ba_engine *e, int32_t horizon, const double *newX, double *out);
/* state(): T doubles of one chain; level sigsq; level suf (n, sumsq) */
int ba_ss_get_state(ba_engine *e, int64_t chain, double *state,
                    double *level_sigsq, double *level_n, double *level_sumsq);
int ba_ss_set_level_sigsq(ba_engine *e, int64_t chain, double sigsq);
/* per-chain regression sufficient statistics left by impute_state */
int ba_ss_get_chain_suf(ba_engine *e, int64_t chain, double *xty, double *yty,
                        double *n);

/* ---- several devices behind one handle (north-star: "C++ host code ...
 * chains shard across the GPUs of one node with a single RCCL gather") --------------
 * A group is one engine per entry of `devices` in ONE process: engine i lives on HIP
 * device devices[i] with its own stream and owns the global chains
 * [i * chains_per_device, (i + 1) * chains_per_device), so a chain's draws do not
 * depend on the device that runs it.  Nothing on the sampling path crosses devices;
 * the two collectives go through librccl directly (loaded on demand; xGMI between the
 * devices of a node): ba_group_build_suf_from_xy = rows of X sharded over the devices,
 * local f64-MFMA syrk, ONE ncclAllReduce of (X'X | X'y | y'y, sum y | sum x), every
 * engine installs the bitwise identical total; ba_group_get_summaries = ONE
 * ncclAllGather of the per-device summary blocks, summed on the host.  A device list
 * that repeats ONE device is allowed (no collective: blocks are summed on the device);
 * a mixed list is not.  Everything that is per engine (priors beyond the common ones
 * below, options, state accessors, look-ahead, the other samplers) is reached through
 * ba_group_engine(g, i) with the single-engine entry points; ba_group_locate maps a
 * global chain id to (engine, local chain).  Errors: the BA_E_* codes, message from
 * ba_group_last_error().  New: the reference has no multi-chain or multi-device
 * driver. */
typedef struct ba_group ba_group;
const char *ba_group_last_error(void);
/* 1 when librccl can be loaded and exports what a multi-device group calls */
int32_t ba_group_rccl_available(void);
int ba_group_create(const int32_t *devices, int32_t ndevices, int32_t chains_per_device,
                    uint64_t seed, ba_group **out);
void ba_group_destroy(ba_group *g);
int32_t ba_group_size(const ba_group *g);
ba_engine *ba_group_engine(ba_group *g, int32_t i);
int ba_group_locate(const ba_group *g, int64_t global_chain, int32_t *engine_index,
                    int64_t *local_chain);
/* NeRegSuf(X, y) for a host matrix (n x p column-major): see above */
int ba_group_build_suf_from_xy(ba_group *g, int64_t n, int32_t p, const double *X,
                               const double *y);
/* ba_set_slab + ba_set_spike + ba_set_sigma_prior on every engine */
int ba_group_set_priors(ba_group *g, const double *prior_mean,
                        const double *unscaled_prior_precision,
                        const double *prior_inclusion_probabilities, int64_t max_model_size,
                        double prior_df, double sigma_guess, double sigma_upper_limit);
/* ba_set_state(e, -1, ...) on every engine */
int ba_group_set_state(ba_group *g, const uint8_t *gamma, const double *beta, double sigsq);
/* ba_sweep on every engine (asynchronous: the devices run side by side); ba_sync */
int ba_group_sweep(ba_group *g, int32_t nsweeps);
/* fn(engine, arg) on every engine of the group, in order; stops at the first error.  What
 * is the same on every device -- the bsts / logit / Poisson data (replicated), priors, the
 * state models, options, the look-ahead -- is set this way with the single-engine entry
 * points; the sweeps of those samplers go out on every device before any is waited for. */
int ba_group_call(ba_group *g, int (*fn)(ba_engine *, void *), void *arg);
int ba_group_ss_sweep(ba_group *g, int32_t nsweeps);
int ba_group_ss_draw_next(ba_group *g);
int ba_group_logit_sweep(ba_group *g, int32_t nsweeps);
int ba_group_poisson_sweep(ba_group *g, int32_t nsweeps);
int ba_group_sync(ba_group *g);
int ba_group_reset_summaries(ba_group *g);
/* whole-job summaries, laid out as ba_get_summaries; blocks (may be NULL) receives the
 * gathered per-device blocks, ba_group_size(g) x (3p + 16) doubles */
int ba_group_get_summaries(ba_group *g, double *inclusion_count, double *beta_sum,
                           double *beta_sumsq, double *scalars, double *blocks);

#ifdef __cplusplus
}
#endif
#endif /* BOOM_AMD_H */
