// TEST INFRASTRUCTURE -- not part of the shipped product.
//
// C-ABI driver around the *unmodified* BOOM reference at /root/reference.  It
// is compiled only in the build container (see oracle/Makefile) into
// oracle/_ref/libboomref.so and is used for two things:
//   1. to pin the clean-room restatement in oracle/boom_oracle.c (same seeds,
//      std::mt19937_64 stream => same draws), and
//   2. to generate the golden fixtures under tests/golden/ (see
//      tests/golden/make_golden.py).
// This file is our own code: it only *calls* the reference's public classes.
// It never ships to the GPU box in source form that is needed at run time;
// nothing in boom_amd/ links it.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "Models/Glm/PoissonRegressionModel.hpp"
#include "Models/Glm/PosteriorSamplers/PoissonRegressionSpikeSlabSampler.hpp"
#include "Models/Glm/PosteriorSamplers/NormalMixtureApproximation.hpp"
#include "Models/Glm/PosteriorSamplers/poisson_mixture_approximation_table.hpp"
#include "LinAlg/Cholesky.hpp"
#include "LinAlg/Matrix.hpp"
#include "LinAlg/Selector.hpp"
#include "LinAlg/SpdMatrix.hpp"
#include "LinAlg/Vector.hpp"
#include "Models/ChisqModel.hpp"
#include "Models/Glm/PosteriorSamplers/AdaptiveSpikeSlabRegressionSampler.hpp"
#include "Models/Glm/PosteriorSamplers/BregVsSampler.hpp"
#include "Models/Glm/PosteriorSamplers/SpikeSlabSampler.hpp"
#include "Models/Glm/BinomialProbitModel.hpp"
#include "Models/Glm/BinomialLogitModel.hpp"
#include "Models/Glm/PosteriorSamplers/BinomialLogitSpikeSlabSampler.hpp"
#include "Models/Glm/PosteriorSamplers/BinomialProbitSpikeSlabSampler.hpp"
#include "Models/Glm/WeightedRegressionModel.hpp"
#include "Models/MvnModel.hpp"
#include "Models/Glm/RegressionModel.hpp"
#include "Models/Glm/VariableSelectionPrior.hpp"
#include "Models/MvnGivenScalarSigma.hpp"
#include "Models/PosteriorSamplers/ZeroMeanGaussianConjSampler.hpp"
#include "Models/StateSpace/PosteriorSamplers/StateSpacePosteriorSampler.hpp"
#include "Models/StateSpace/StateModels/ArStateModel.hpp"
#include "Models/StateSpace/StateModels/LocalLevelStateModel.hpp"
#include "Models/TimeSeries/PosteriorSamplers/ArPosteriorSampler.hpp"
#include "Models/StateSpace/StateModels/LocalLinearTrend.hpp"
#include "Models/StateSpace/StateModels/SeasonalStateModel.hpp"
#include "Models/StateSpace/StateModels/StaticInterceptStateModel.hpp"
#include "Models/StateSpace/StateModels/SemilocalLinearTrend.hpp"
#include "Models/TimeSeries/NonzeroMeanAr1Model.hpp"
#include "Models/TimeSeries/PosteriorSamplers/NonzeroMeanAr1Sampler.hpp"
#include "Models/GaussianModel.hpp"
#include "Models/StateSpace/StateModels/TrigStateModel.hpp"
#include "Models/PosteriorSamplers/ZeroMeanMvnIndependenceSampler.hpp"
#include "Models/StateSpace/StateSpaceRegressionModel.hpp"
#include "cpputil/shuffle.hpp"
#include "distributions.hpp"
#include "distributions/rng.hpp"
#include "distributions/trun_gamma.hpp"

using namespace BOOM;

static std::string g_err;

#define REF_TRY try {
#define REF_CATCH                                  \
  }                                                \
  catch (std::exception & e) {                     \
    g_err = e.what();                              \
    return 1;                                      \
  }                                                \
  catch (...) {                                    \
    g_err = "unknown exception";                   \
    return 2;                                      \
  }                                                \
  return 0;

static Matrix make_matrix(int nr, int nc, const double *colmajor) {
  Matrix m(nr, nc);
  std::memcpy(m.data(), colmajor, sizeof(double) * nr * nc);
  return m;
}
static SpdMatrix make_spd(int n, const double *colmajor) {
  SpdMatrix m(n);
  std::memcpy(m.data(), colmajor, sizeof(double) * n * n);
  return m;
}
static Vector make_vector(int n, const double *x) {
  Vector v(n);
  std::memcpy(v.data(), x, sizeof(double) * n);
  return v;
}

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

// ---------------------------------------------------------------- RNG KATs
int ref_rng_uniform(uint64_t seed, int n, double *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = rng();
  REF_CATCH
}

int ref_seed_rng(uint64_t seed, int n, uint64_t *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = seed_rng(rng);
  REF_CATCH
}

int ref_rng_norm(uint64_t seed, int n, double *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = rnorm_mt(rng, 0, 1);
  REF_CATCH
}

int ref_rng_exp(uint64_t seed, int n, double *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = rexp_mt(rng, 1.0);
  REF_CATCH
}

// BOOM::rgamma_mt(rng, a, b) with b a *rate*.
int ref_rng_gamma(uint64_t seed, double a, double b, int n, double *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = rgamma_mt(rng, a, b);
  REF_CATCH
}

int ref_rng_trun_gamma(uint64_t seed, double a, double b, double cut, int n,
                       double *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = rtrun_gamma_mt(rng, a, b, cut);
  REF_CATCH
}

int ref_rng_random_int(uint64_t seed, int lo, int hi, int n, int *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = random_int_mt(rng, lo, hi);
  REF_CATCH
}

// A persistent index vector shuffled nrep times in place (as BregVsSampler's
// `indx` member is); out is nrep x p.
int ref_rng_shuffle(uint64_t seed, int p, int nrep, int *out) {
  REF_TRY
  RNG rng(seed);
  std::vector<int> v(p);
  for (int i = 0; i < p; ++i) v[i] = i;
  for (int r = 0; r < nrep; ++r) {
    shuffle(v, rng);
    for (int i = 0; i < p; ++i) out[r * p + i] = v[i];
  }
  REF_CATCH
}

int ref_rng_rmulti(uint64_t seed, int k, const double *prob, int n, int *out) {
  REF_TRY
  RNG rng(seed);
  Vector pr = make_vector(k, prob);
  for (int i = 0; i < n; ++i) out[i] = rmulti_mt(rng, pr);
  REF_CATCH
}

// ---------------------------------------------------------- LinAlg KATs
int ref_spd_logdet(int n, const double *A, double *out, int *ok) {
  REF_TRY
  SpdMatrix S = make_spd(n, A);
  bool good = true;
  *out = S.logdet(good);
  *ok = good;
  REF_CATCH
}

int ref_spd_chol(int n, const double *A, double *L, int *ok) {
  REF_TRY
  SpdMatrix S = make_spd(n, A);
  bool good = true;
  Matrix l = S.chol(good);
  *ok = good;
  if (good) std::memcpy(L, l.data(), sizeof(double) * n * n);
  REF_CATCH
}

int ref_spd_solve(int n, const double *A, const double *rhs, double *out,
                  int *ok) {
  REF_TRY
  SpdMatrix S = make_spd(n, A);
  bool good = true;
  Vector ans = S.solve(make_vector(n, rhs), good);
  *ok = good;
  std::memcpy(out, ans.data(), sizeof(double) * n);
  REF_CATCH
}

int ref_spd_mdist(int n, const double *A, const double *x, double *out) {
  REF_TRY
  SpdMatrix S = make_spd(n, A);
  *out = S.Mdist(make_vector(n, x));
  REF_CATCH
}

// ------------------------------------------------------- sufficient stats
int ref_neregsuf(int n, int p, const double *X, const double *y, double *xtx,
                 double *xty, double *yty, double *ybar, double *xbar) {
  REF_TRY
  Matrix Xm = make_matrix(n, p, X);
  Vector yv = make_vector(n, y);
  NeRegSuf suf(Xm, yv);
  SpdMatrix S = suf.xtx();
  std::memcpy(xtx, S.data(), sizeof(double) * p * p);
  Vector s = suf.xty();
  std::memcpy(xty, s.data(), sizeof(double) * p);
  *yty = suf.yty();
  *ybar = suf.ybar();
  Vector xb = suf.xbar();
  std::memcpy(xbar, xb.data(), sizeof(double) * p);
  REF_CATCH
}

// ------------------------------------------------------------- SSVS runs
struct RefSsvsOptions {
  int64_t max_model_size;    // < 0: no limit
  double sigma_upper_limit;  // +inf: none
  double swap_threshold;     // >= 1 disables the swap move
  int max_flips;             // < 0: p
  int draw_beta;             // NOTE reference bug: allow_* sets the flag false
  int draw_sigma;
};

static void record(const RegressionModel &model, int p, int sweep,
                   uint8_t *out_gamma, double *out_beta, double *out_sigsq) {
  const Selector &inc(model.coef().inc());
  const Vector &beta(model.Beta());
  for (int j = 0; j < p; ++j) {
    out_gamma[(size_t)sweep * p + j] = inc[j] ? 1 : 0;
    out_beta[(size_t)sweep * p + j] = beta[j];
  }
  out_sigsq[sweep] = model.sigsq();
}

static void apply_options(BregVsSampler &sampler, const Ptr<VariableSelectionPrior> &spike,
                          const RefSsvsOptions *opt) {
  if (opt->max_model_size >= 0) spike->set_max_model_size(opt->max_model_size);
  if (std::isfinite(opt->sigma_upper_limit)) {
    sampler.set_sigma_upper_limit(opt->sigma_upper_limit);
  }
  sampler.set_correlation_swap_threshold(opt->swap_threshold);
  if (opt->max_flips >= 0) sampler.limit_model_selection(opt->max_flips);
  if (!opt->draw_beta) sampler.suppress_beta_draw();
  if (!opt->draw_sigma) sampler.suppress_sigma_draw();
}

// Raw-prior run (ctor #5: model objects).  Data are given either as (X, y)
// [X != NULL] or as sufficient statistics.  GlobalRng is seeded with `seed`
// and the sampler takes its private seed from it, exactly as the R / Python
// wrappers do (spike_slab_wrapper.cc:201-254).
int ref_ssvs_run(int n, int p, const double *X, const double *y,
                 const double *xtx, const double *xty, double yty, double ybar,
                 const double *xbar, const double *prior_mean,
                 const double *ominv, double prior_df, double sigma_guess,
                 const double *pi, const RefSsvsOptions *opt, uint64_t seed,
                 const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                 double *out_beta, double *out_sigsq) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  Ptr<RegressionModel> model;
  if (X) {
    model = new RegressionModel(make_matrix(n, p, X), make_vector(n, y), false);
  } else {
    NEW(NeRegSuf, suf)(make_spd(p, xtx), make_vector(p, xty), yty, (double)n,
                       ybar, make_vector(p, xbar));
    model = new RegressionModel(Ptr<RegSuf>(suf));
  }
  NEW(MvnGivenScalarSigma, slab)(make_vector(p, prior_mean), make_spd(p, ominv),
                                 model->Sigsq_prm());
  NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  NEW(BregVsSampler, sampler)(model.get(), slab, siginv_prior, spike);
  apply_options(*sampler, spike, opt);
  model->set_method(sampler);
  model->coef().drop_all();
  for (int j = 0; j < p; ++j) {
    if (init_gamma[j]) model->coef().add(j);
  }
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    record(*model, p, i, out_gamma, out_beta, out_sigsq);
  }
  REF_CATCH
}

// ---- AdaptiveSpikeSlabRegressionSampler (what lm.spike uses for p > 100)
//      Models/Glm/PosteriorSamplers/AdaptiveSpikeSlabRegressionSampler.cpp:62-225
int ref_adaptive_run(int n, int p, const double *xtx, const double *xty, double yty,
                     double ybar, const double *xbar, const double *prior_mean,
                     const double *ominv, double prior_df, double sigma_guess,
                     const double *pi, int64_t max_model_size,
                     double sigma_upper_limit, int max_flips, double step_size,
                     double target_acceptance_rate, uint64_t seed,
                     const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                     double *out_beta, double *out_sigsq) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  NEW(NeRegSuf, suf)(make_spd(p, xtx), make_vector(p, xty), yty, (double)n, ybar,
                     make_vector(p, xbar));
  Ptr<RegressionModel> model(new RegressionModel(Ptr<RegSuf>(suf)));
  NEW(MvnGivenScalarSigma, slab)(make_vector(p, prior_mean), make_spd(p, ominv),
                                 model->Sigsq_prm());
  NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
  NEW(AdaptiveSpikeSlabRegressionSampler, sampler)(model.get(), slab, siginv_prior, spike);
  if (std::isfinite(sigma_upper_limit)) sampler->set_sigma_upper_limit(sigma_upper_limit);
  if (max_flips >= 0) sampler->limit_model_selection(max_flips);
  if (step_size > 0) sampler->set_step_size(step_size);
  if (target_acceptance_rate > 0) sampler->set_target_acceptance_rate(target_acceptance_rate);
  model->set_method(sampler);
  model->coef().drop_all();
  for (int j = 0; j < p; ++j) {
    if (init_gamma[j]) model->coef().add(j);
  }
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    record(*model, p, i, out_gamma, out_beta, out_sigsq);
  }
  REF_CATCH
}

// Convenience-ctor run:
//   which == 1: (model, prior_nobs=a0, expected_rsq=a1, expected_model_size=a2,
//                first_term_is_intercept=flag)            BregVsSampler.cpp:48-85
//   which == 2: (model, prior_sigma_nobs=a0, prior_sigma_guess=a1,
//                prior_beta_nobs=a2, diagonal_shrinkage=a3,
//                prior_inclusion_probability=a4, force_intercept=flag)  :87-142
int ref_ssvs_run_ctor(int which, int n, int p, const double *X, const double *y,
                      double a0, double a1, double a2, double a3, double a4,
                      int flag, const RefSsvsOptions *opt, uint64_t seed,
                      const uint8_t *init_gamma, int nsweeps,
                      uint8_t *out_gamma, double *out_beta, double *out_sigsq) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  NEW(RegressionModel, model)(make_matrix(n, p, X), make_vector(n, y), false);
  Ptr<BregVsSampler> sampler;
  if (which == 1) {
    sampler = new BregVsSampler(model.get(), a0, a1, a2, flag != 0);
  } else {
    sampler = new BregVsSampler(model.get(), a0, a1, a2, a3, a4, flag != 0);
  }
  RefSsvsOptions o = *opt;
  o.max_model_size = -1;
  if (std::isfinite(o.sigma_upper_limit)) {
    sampler->set_sigma_upper_limit(o.sigma_upper_limit);
  }
  sampler->set_correlation_swap_threshold(o.swap_threshold);
  if (o.max_flips >= 0) sampler->limit_model_selection(o.max_flips);
  model->set_method(sampler);
  model->coef().drop_all();
  for (int j = 0; j < p; ++j) {
    if (init_gamma[j]) model->coef().add(j);
  }
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    record(*model, p, i, out_gamma, out_beta, out_sigsq);
  }
  REF_CATCH
}

// log_model_prob of an arbitrary inclusion vector (pins a5/a6 directly).
int ref_ssvs_log_model_prob(int n, int p, const double *xtx, const double *xty,
                            double yty, double ybar, const double *xbar,
                            const double *prior_mean, const double *ominv,
                            double prior_df, double sigma_guess,
                            const double *pi, int64_t max_model_size,
                            int ngamma, const uint8_t *gammas, double *out) {
  REF_TRY
  GlobalRng::rng.seed(1);
  NEW(NeRegSuf, suf)(make_spd(p, xtx), make_vector(p, xty), yty, (double)n,
                     ybar, make_vector(p, xbar));
  NEW(RegressionModel, model)(Ptr<RegSuf>(suf));
  NEW(MvnGivenScalarSigma, slab)(make_vector(p, prior_mean), make_spd(p, ominv),
                                 model->Sigsq_prm());
  NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
  NEW(BregVsSampler, sampler)(model.get(), slab, siginv_prior, spike);
  for (int g = 0; g < ngamma; ++g) {
    Selector inc(p, false);
    for (int j = 0; j < p; ++j) {
      if (gammas[(size_t)g * p + j]) inc.add(j);
    }
    out[g] = sampler->log_model_prob(inc);
  }
  REF_CATCH
}

// BregVsSampler::logpri() (BregVsSampler.cpp:380-393) at given states.
int ref_ssvs_logpri(int n, int p, const double *xtx, const double *xty,
                    double yty, double ybar, const double *xbar,
                    const double *prior_mean, const double *ominv,
                    double prior_df, double sigma_guess, const double *pi,
                    int64_t max_model_size, int nstates, const uint8_t *gammas,
                    const double *betas, const double *sigsqs, double *out) {
  REF_TRY
  GlobalRng::rng.seed(1);
  NEW(NeRegSuf, suf)(make_spd(p, xtx), make_vector(p, xty), yty, (double)n,
                     ybar, make_vector(p, xbar));
  NEW(RegressionModel, model)(Ptr<RegSuf>(suf));
  NEW(MvnGivenScalarSigma, slab)(make_vector(p, prior_mean), make_spd(p, ominv),
                                 model->Sigsq_prm());
  NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
  NEW(BregVsSampler, sampler)(model.get(), slab, siginv_prior, spike);
  for (int s = 0; s < nstates; ++s) {
    Selector inc(p, false);
    for (int j = 0; j < p; ++j) {
      if (gammas[(size_t)s * p + j]) inc.add(j);
    }
    model->coef().set_inc(inc);
    model->set_Beta(make_vector(p, betas + (size_t)s * p));
    model->set_sigsq(sigsqs[s]);
    out[s] = sampler->logpri();
  }
  REF_CATCH
}

// ------------------------------------------------------ state-space runs
// bsts "local level + regression" (SURVEY 3.3): StateSpaceRegressionModel with
// one LocalLevelStateModel, BregVsSampler on the observation model,
// ZeroMeanGaussianConjSampler on the level variance, StateSpacePosteriorSampler
// on top.  Samplers are constructed in that order (each takes its private seed
// from GlobalRng at construction, PosteriorSampler.cpp:34-35).
struct RefSsOptions {
  double level_df;
  double level_sigma_guess;
  double level_sigma_upper_limit;  // +inf: none
  double initial_state_mean;
  double initial_state_variance;
  double initial_level_sigma;
};

int ref_ss_run(int T, int p, const double *y, const double *X,
               const uint8_t *observed, const double *prior_mean,
               const double *ominv, double prior_df, double sigma_guess,
               const double *pi, const RefSsvsOptions *opt,
               const RefSsOptions *ss, uint64_t seed, const uint8_t *init_gamma,
               int nsweeps, uint8_t *out_gamma, double *out_beta,
               double *out_sigsq, double *out_level_sigsq, double *out_state) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  std::vector<bool> obs;
  if (observed) {
    obs.resize(T);
    for (int t = 0; t < T; ++t) obs[t] = observed[t] != 0;
  }
  NEW(StateSpaceRegressionModel, model)(make_vector(T, y), make_matrix(T, p, X),
                                        obs);
  RegressionModel *reg = model->observation_model();
  NEW(MvnGivenScalarSigma, slab)(make_vector(p, prior_mean), make_spd(p, ominv),
                                 reg->Sigsq_prm());
  NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  NEW(BregVsSampler, reg_sampler)(reg, slab, siginv_prior, spike);
  apply_options(*reg_sampler, spike, opt);
  reg->set_method(reg_sampler);
  reg->coef().drop_all();
  for (int j = 0; j < p; ++j) {
    if (init_gamma[j]) reg->coef().add(j);
  }

  NEW(LocalLevelStateModel, level)(ss->initial_level_sigma);
  NEW(ZeroMeanGaussianConjSampler, level_sampler)(level.get(), ss->level_df,
                                                 ss->level_sigma_guess);
  if (std::isfinite(ss->level_sigma_upper_limit)) {
    level_sampler->set_sigma_upper_limit(ss->level_sigma_upper_limit);
  }
  level->set_method(level_sampler);
  level->set_initial_state_mean(ss->initial_state_mean);
  level->set_initial_state_variance(ss->initial_state_variance);
  model->add_state(level);

  NEW(StateSpacePosteriorSampler, sampler)(model.get());
  model->set_method(sampler);

  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    record(*reg, p, i, out_gamma, out_beta, out_sigsq);
    out_level_sigsq[i] = level->sigsq();
    const Matrix &state(model->state());
    for (int t = 0; t < T; ++t) out_state[(size_t)i * T + t] = state(0, t);
  }
  REF_CATCH
}

// Structural time series: regression + trend (local level or
// local linear trend with independent variance samplers, as bsts builds it) +
// optional seasonal state.  Three-element arrays: level, slope, seasonal.
// ar_lags > 0 adds an ArStateModel(ar_lags) + ArPosteriorSampler after the trend /
// seasonal blocks: ar[] = {prior df, prior sigma guess, sigma upper limit,
// initial sigma}, initial phi ar_phi0[lags]; the block's initial state moments
// follow the others in initial_state_mean / _variance.  out_ar: nsweeps x (lags + 1)
// = phi, sigsq.
static int ssm_run_impl(int T, int p, const double *y, const double *X,
                const uint8_t *observed, const double *prior_mean,
                const double *ominv, double prior_df, double sigma_guess,
                const double *pi, const RefSsvsOptions *opt, int trend,
                int nseasons, const double *var_df, const double *var_sigma_guess,
                const double *var_sigma_upper_limit, const double *var_initial_sigma,
                const double *initial_state_mean,
                const double *initial_state_variance, int ar_lags, const double *ar,
                const double *ar_phi0, uint64_t seed,
                const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                double *out_beta, double *out_sigsq, double *out_variances,
                double *out_state, double *out_ar) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  std::vector<bool> obs;
  if (observed) {
    obs.resize(T);
    for (int t = 0; t < T; ++t) obs[t] = observed[t] != 0;
  }
  NEW(StateSpaceRegressionModel, model)(make_vector(T, y), make_matrix(T, p, X),
                                        obs);
  RegressionModel *reg = model->observation_model();
  NEW(MvnGivenScalarSigma, slab)(make_vector(p, prior_mean), make_spd(p, ominv),
                                 reg->Sigsq_prm());
  NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  NEW(BregVsSampler, reg_sampler)(reg, slab, siginv_prior, spike);
  apply_options(*reg_sampler, spike, opt);
  reg->set_method(reg_sampler);
  reg->coef().drop_all();
  for (int j = 0; j < p; ++j) {
    if (init_gamma[j]) reg->coef().add(j);
  }

  Ptr<LocalLevelStateModel> level;
  Ptr<LocalLinearTrendStateModel> llt;
  Ptr<SeasonalStateModel> seasonal;
  if (trend == 1) {
    level = new LocalLevelStateModel(var_initial_sigma[0]);
    NEW(ZeroMeanGaussianConjSampler, level_sampler)(level.get(), var_df[0],
                                                   var_sigma_guess[0]);
    if (std::isfinite(var_sigma_upper_limit[0]))
      level_sampler->set_sigma_upper_limit(var_sigma_upper_limit[0]);
    level->set_method(level_sampler);
    level->set_initial_state_mean(initial_state_mean[0]);
    level->set_initial_state_variance(initial_state_variance[0]);
    model->add_state(level);
  } else {
    llt = new LocalLinearTrendStateModel;
    SpdMatrix Sigma(2, 0.0);
    Sigma(0, 0) = var_initial_sigma[0] * var_initial_sigma[0];
    Sigma(1, 1) = var_initial_sigma[1] * var_initial_sigma[1];
    llt->set_Sigma(Sigma);
    for (int i = 0; i < 2; ++i) {
      NEW(ZeroMeanMvnIndependenceSampler, s)(llt.get(), var_df[i],
                                             var_sigma_guess[i], i);
      if (std::isfinite(var_sigma_upper_limit[i]))
        s->set_sigma_upper_limit(var_sigma_upper_limit[i]);
      llt->set_method(s);
    }
    Vector a0(2);
    SpdMatrix P0(2, 0.0);
    for (int i = 0; i < 2; ++i) {
      a0[i] = initial_state_mean[i];
      P0(i, i) = initial_state_variance[i];
    }
    llt->set_initial_state_mean(a0);
    llt->set_initial_state_variance(P0);
    model->add_state(llt);
  }
  if (nseasons > 0) {
    seasonal = new SeasonalStateModel(nseasons, 1);
    seasonal->set_sigsq(var_initial_sigma[2] * var_initial_sigma[2]);
    NEW(ZeroMeanGaussianConjSampler, seas_sampler)(seasonal.get(), var_df[2],
                                                  var_sigma_guess[2]);
    if (std::isfinite(var_sigma_upper_limit[2]))
      seas_sampler->set_sigma_upper_limit(var_sigma_upper_limit[2]);
    seasonal->set_method(seas_sampler);
    const int n = nseasons - 1;
    Vector a0(n);
    SpdMatrix P0(n, 0.0);
    for (int i = 0; i < n; ++i) {
      a0[i] = initial_state_mean[trend + i];
      P0(i, i) = initial_state_variance[trend + i];
    }
    seasonal->set_initial_state_mean(a0);
    seasonal->set_initial_state_variance(P0);
    model->add_state(seasonal);
  }

  Ptr<ArStateModel> arm;
  const int m0 = trend + (nseasons > 0 ? nseasons - 1 : 0);
  if (ar_lags > 0) {
    arm = new ArStateModel(ar_lags);
    arm->set_phi(make_vector(ar_lags, ar_phi0));
    arm->set_sigma(ar[3]);
    NEW(ChisqModel, ar_prior)(ar[0], ar[1]);
    NEW(ArPosteriorSampler, ar_sampler)(arm.get(), ar_prior);
    if (std::isfinite(ar[2])) ar_sampler->set_sigma_upper_limit(ar[2]);
    arm->set_method(ar_sampler);
    Vector a0(ar_lags);
    SpdMatrix P0(ar_lags, 0.0);
    for (int i = 0; i < ar_lags; ++i) {
      a0[i] = initial_state_mean[m0 + i];
      P0(i, i) = initial_state_variance[m0 + i];
    }
    arm->set_initial_state_mean(a0);
    arm->set_initial_state_variance(P0);
    model->add_state(arm);
  }

  NEW(StateSpacePosteriorSampler, sampler)(model.get());
  model->set_method(sampler);
  const int m = m0 + ar_lags;
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    record(*reg, p, i, out_gamma, out_beta, out_sigsq);
    out_variances[3 * i + 0] = trend == 1 ? level->sigsq() : llt->Sigma()(0, 0);
    out_variances[3 * i + 1] = trend == 1 ? 0.0 : llt->Sigma()(1, 1);
    out_variances[3 * i + 2] = nseasons > 0 ? seasonal->sigsq() : 0.0;
    if (ar_lags > 0) {
      for (int j = 0; j < ar_lags; ++j) out_ar[(size_t)i * (ar_lags + 1) + j] = arm->phi()[j];
      out_ar[(size_t)i * (ar_lags + 1) + ar_lags] = arm->sigsq();
    }
    const Matrix &state(model->state());
    for (int t = 0; t < T; ++t)
      for (int j = 0; j < m; ++j) out_state[((size_t)i * T + t) * m + j] = state(j, t);
  }
  REF_CATCH
}

int ref_ssm_run(int T, int p, const double *y, const double *X,
                const uint8_t *observed, const double *prior_mean,
                const double *ominv, double prior_df, double sigma_guess,
                const double *pi, const RefSsvsOptions *opt, int trend,
                int nseasons, const double *var_df, const double *var_sigma_guess,
                const double *var_sigma_upper_limit, const double *var_initial_sigma,
                const double *initial_state_mean,
                const double *initial_state_variance, uint64_t seed,
                const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                double *out_beta, double *out_sigsq, double *out_variances,
                double *out_state) {
  return ssm_run_impl(T, p, y, X, observed, prior_mean, ominv, prior_df, sigma_guess, pi, opt,
                      trend, nseasons, var_df, var_sigma_guess, var_sigma_upper_limit,
                      var_initial_sigma, initial_state_mean, initial_state_variance, 0, nullptr,
                      nullptr, seed, init_gamma, nsweeps, out_gamma, out_beta, out_sigsq,
                      out_variances, out_state, nullptr);
}

// ArModel::check_stationary (the quick bound, then Jenkins-Traub roots)
int ref_ar_check_stationary(int lags, const double *phi) {
  try {
    return ArModel::check_stationary(make_vector(lags, phi)) ? 1 : 0;
  } catch (...) {
    return -1;
  }
}

// the same with an ArStateModel block (see ssm_run_impl)
int ref_ssm_ar_run(int T, int p, const double *y, const double *X,
                   const uint8_t *observed, const double *prior_mean,
                   const double *ominv, double prior_df, double sigma_guess,
                   const double *pi, const RefSsvsOptions *opt, int trend,
                   int nseasons, const double *var_df, const double *var_sigma_guess,
                   const double *var_sigma_upper_limit, const double *var_initial_sigma,
                   const double *initial_state_mean,
                   const double *initial_state_variance, int ar_lags, const double *ar,
                   const double *ar_phi0, uint64_t seed,
                   const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                   double *out_beta, double *out_sigsq, double *out_variances,
                   double *out_state, double *out_ar) {
  return ssm_run_impl(T, p, y, X, observed, prior_mean, ominv, prior_df, sigma_guess, pi, opt,
                      trend, nseasons, var_df, var_sigma_guess, var_sigma_upper_limit,
                      var_initial_sigma, initial_state_mean, initial_state_variance, ar_lags, ar,
                      ar_phi0, seed, init_gamma, nsweeps, out_gamma, out_beta, out_sigsq,
                      out_variances, out_state, out_ar);
}

// One isolated impute_state() with fixed parameters: pins a14-a18 (filter,
// simulation, disturbance smoother, mean correction, sufficient statistics)
// without any parameter draw in between.  Outputs the drawn state and the
// regression / level sufficient statistics left behind.
int ref_ss_impute_state(int T, int p, const double *y, const double *X,
                        const uint8_t *observed, const double *beta,
                        const uint8_t *gamma, double sigsq_obs,
                        double sigsq_level, double a0, double P0,
                        uint64_t seed, double *out_state, double *out_xty,
                        double *out_yty, double *out_n, double *out_level_sumsq,
                        double *out_level_n) {
  REF_TRY
  std::vector<bool> obs;
  if (observed) {
    obs.resize(T);
    for (int t = 0; t < T; ++t) obs[t] = observed[t] != 0;
  }
  NEW(StateSpaceRegressionModel, model)(make_vector(T, y), make_matrix(T, p, X),
                                        obs);
  RegressionModel *reg = model->observation_model();
  reg->coef().drop_all();
  Vector b(p, 0.0);
  for (int j = 0; j < p; ++j) {
    if (gamma[j]) {
      reg->coef().add(j);
      b[j] = beta[j];
    }
  }
  reg->coef().set_Beta(b);
  reg->set_sigsq(sigsq_obs);
  NEW(LocalLevelStateModel, level)(std::sqrt(sigsq_level));
  level->set_initial_state_mean(a0);
  level->set_initial_state_variance(P0);
  model->add_state(level);
  RNG rng(seed);
  model->impute_state(rng);
  const Matrix &state(model->state());
  for (int t = 0; t < T; ++t) out_state[t] = state(0, t);
  Vector xty = reg->suf()->xty();
  std::memcpy(out_xty, xty.data(), sizeof(double) * p);
  *out_yty = reg->suf()->yty();
  *out_n = reg->suf()->n();
  *out_level_sumsq = level->suf()->sumsq();
  *out_level_n = level->suf()->n();
  REF_CATCH
}

// simulate_forecast (StateSpaceRegressionModel.cpp:214-219, :256-278): the draw
// from the predictive distribution of the next `horizon` observations given
// the parameters and the final state of one MCMC draw (what bsts' predict does
// per saved draw)
int ref_ss_forecast(int T, int p, const double *y, const double *X,
                    const double *beta, const uint8_t *gamma, double sigsq_obs,
                    double sigsq_level, double final_state, int horizon,
                    const double *newX, uint64_t seed, double *out) {
  REF_TRY
  NEW(StateSpaceRegressionModel, model)(make_vector(T, y), make_matrix(T, p, X),
                                        std::vector<bool>());
  RegressionModel *reg = model->observation_model();
  reg->coef().drop_all();
  Vector b(p, 0.0);
  for (int j = 0; j < p; ++j) {
    if (gamma[j]) {
      reg->coef().add(j);
      b[j] = beta[j];
    }
  }
  reg->coef().set_Beta(b);
  reg->set_sigsq(sigsq_obs);
  NEW(LocalLevelStateModel, level)(std::sqrt(sigsq_level));
  model->add_state(level);
  RNG rng(seed);
  Vector fs(1, final_state);
  Vector ans = model->simulate_forecast(rng, make_matrix(horizon, p, newX), fs);
  for (int i = 0; i < horizon; ++i) out[i] = ans[i];
  REF_CATCH
}

// simulate_forecast of the structural model: fixed parameters and final state
int ref_ssm_forecast(int T, int p, const double *y, const double *X, const double *beta,
                     const uint8_t *gamma, double sigsq_obs, int trend, int nseasons,
                     const double *sigsq, const double *final_state, int horizon,
                     const double *newX, uint64_t seed, double *out) {
  REF_TRY
  NEW(StateSpaceRegressionModel, model)(make_vector(T, y), make_matrix(T, p, X),
                                        std::vector<bool>());
  RegressionModel *reg = model->observation_model();
  reg->coef().drop_all();
  Vector b(p, 0.0);
  for (int j = 0; j < p; ++j) {
    if (gamma[j]) {
      reg->coef().add(j);
      b[j] = beta[j];
    }
  }
  reg->coef().set_Beta(b);
  reg->set_sigsq(sigsq_obs);
  if (trend == 1) {
    NEW(LocalLevelStateModel, level)(std::sqrt(sigsq[0]));
    model->add_state(level);
  } else {
    NEW(LocalLinearTrendStateModel, llt)();
    SpdMatrix Sigma(2, 0.0);
    Sigma(0, 0) = sigsq[0];
    Sigma(1, 1) = sigsq[1];
    llt->set_Sigma(Sigma);
    model->add_state(llt);
  }
  if (nseasons > 0) {
    NEW(SeasonalStateModel, seasonal)(nseasons, 1);
    seasonal->set_sigsq(sigsq[2]);
    seasonal->set_initial_state_mean(Vector(nseasons - 1, 0.0));
    seasonal->set_initial_state_variance(SpdMatrix(nseasons - 1, 1.0));
    model->add_state(seasonal);
  }
  const int m = trend + (nseasons > 0 ? nseasons - 1 : 0);
  RNG rng(seed);
  Vector fs(m);
  for (int i = 0; i < m; ++i) fs[i] = final_state[i];
  Vector ans = model->simulate_forecast(rng, make_matrix(horizon, p, newX), fs);
  for (int i = 0; i < horizon; ++i) out[i] = ans[i];
  REF_CATCH
}

// ---- the general form (round 4): ANY list of state models in any order.
// nblocks blocks; kinds[b] = 1 LocalLevelStateModel, 2 LocalLinearTrendStateModel (one
// ZeroMeanMvnIndependenceSampler per variance), 3 SeasonalStateModel(iparams[3 b],
// iparams[3 b + 1]) with set_time_of_first_observation(iparams[3 b + 2]), 4
// ArStateModel(iparams[3 b]) + ArPosteriorSampler, 5 StaticInterceptStateModel (no
// parameter, no sampler), 6 TrigStateModel(period = phi0[16 b], the iparams[3 b] <= 15
// frequencies phi0[16 b + 1 ..]) with a ZeroMeanGaussianConjSampler on its error
// distribution, as bsts builds it (Interfaces/R/bsts/src/create_state_model.cpp:559-586), 7
// SemilocalLinearTrendStateModel(ZeroMeanGaussianModel level, NonzeroMeanAr1Model slope) with
// the level's ZeroMeanGaussianConjSampler and then the slope's NonzeroMeanAr1Sampler attached
// to the TREND model (create_state_model.cpp:601-672): iparams[3 b + {0, 1}] = force_stationary,
// force_ar1_positive; phi0[16 b + ..] = slope mean prior mu, sigma, slope AR(1) prior mu, sigma,
// initial mu, initial phi; variances (level, slope) in vpar; a0 / P0: level, slope, (ignored).
// vpar[8 b + 4 v + {0, 1, 2, 3}] = prior
// df, prior sigma guess, sigma upper limit (inf: none), initial sigma of variance v of
// block b; phi0[16 b ..] the initial autoregression coefficients; a0 / P0: the blocks'
// initial state means / variances (diagonal) one after the other.
// Outputs per sweep: variances (nblocks x 2), phi (nblocks x 16); the state draw of
// sweeps i with i % state_every == state_every - 1 (T x m each, in order).
struct GeneralState {
  std::vector<Ptr<LocalLevelStateModel>> level;
  std::vector<Ptr<LocalLinearTrendStateModel>> llt;
  std::vector<Ptr<SeasonalStateModel>> seasonal;
  std::vector<Ptr<ArStateModel>> ar;
  std::vector<Ptr<StaticInterceptStateModel>> intercept;
  std::vector<Ptr<TrigStateModel>> trig;
  std::vector<Ptr<SemilocalLinearTrendStateModel>> semilocal;
  std::vector<Ptr<NonzeroMeanAr1Model>> semilocal_slope;
  std::vector<Ptr<ZeroMeanGaussianModel>> semilocal_level;
  std::vector<int> kind, slot;   // per block: which vector, which element
};

static void add_general_state(StateSpaceRegressionModel *model, GeneralState &G, int nblocks,
                              const int *kinds, const int *iparams, const double *vpar,
                              const double *phi0, const double *a0, const double *P0,
                              bool with_samplers) {
  int first = 0;
  for (int b = 0; b < nblocks; ++b) {
    const double *vp = vpar + 8 * b;
    G.kind.push_back(kinds[b]);
    if (kinds[b] == 1) {
      Ptr<LocalLevelStateModel> level(new LocalLevelStateModel(vp[3]));
      if (with_samplers) {
        NEW(ZeroMeanGaussianConjSampler, s)(level.get(), vp[0], vp[1]);
        if (std::isfinite(vp[2])) s->set_sigma_upper_limit(vp[2]);
        level->set_method(s);
      }
      level->set_initial_state_mean(a0[first]);
      level->set_initial_state_variance(P0[first]);
      model->add_state(level);
      G.slot.push_back((int)G.level.size());
      G.level.push_back(level);
      first += 1;
    } else if (kinds[b] == 2) {
      Ptr<LocalLinearTrendStateModel> llt(new LocalLinearTrendStateModel);
      SpdMatrix Sigma(2, 0.0);
      Sigma(0, 0) = vp[3] * vp[3];
      Sigma(1, 1) = vp[7] * vp[7];
      llt->set_Sigma(Sigma);
      if (with_samplers) {
        for (int i = 0; i < 2; ++i) {
          NEW(ZeroMeanMvnIndependenceSampler, s)(llt.get(), vp[4 * i], vp[4 * i + 1], i);
          if (std::isfinite(vp[4 * i + 2])) s->set_sigma_upper_limit(vp[4 * i + 2]);
          llt->set_method(s);
        }
      }
      Vector mean(2);
      SpdMatrix var(2, 0.0);
      for (int i = 0; i < 2; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
      llt->set_initial_state_mean(mean);
      llt->set_initial_state_variance(var);
      model->add_state(llt);
      G.slot.push_back((int)G.llt.size());
      G.llt.push_back(llt);
      first += 2;
    } else if (kinds[b] == 3) {
      const int ns = iparams[3 * b], dur = iparams[3 * b + 1];
      Ptr<SeasonalStateModel> seasonal(new SeasonalStateModel(ns, dur));
      seasonal->set_time_of_first_observation(iparams[3 * b + 2]);
      seasonal->set_sigsq(vp[3] * vp[3]);
      if (with_samplers) {
        NEW(ZeroMeanGaussianConjSampler, s)(seasonal.get(), vp[0], vp[1]);
        if (std::isfinite(vp[2])) s->set_sigma_upper_limit(vp[2]);
        seasonal->set_method(s);
      }
      const int n = ns - 1;
      Vector mean(n);
      SpdMatrix var(n, 0.0);
      for (int i = 0; i < n; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
      seasonal->set_initial_state_mean(mean);
      seasonal->set_initial_state_variance(var);
      model->add_state(seasonal);
      G.slot.push_back((int)G.seasonal.size());
      G.seasonal.push_back(seasonal);
      first += n;
    } else if (kinds[b] == 5) {
      Ptr<StaticInterceptStateModel> icpt(new StaticInterceptStateModel);
      icpt->set_initial_state_mean(a0[first]);
      icpt->set_initial_state_variance(P0[first]);
      model->add_state(icpt);
      G.slot.push_back((int)G.intercept.size());
      G.intercept.push_back(icpt);
      first += 1;
    } else if (kinds[b] == 7) {
      const double *pp = phi0 + 16 * b;
      NEW(ZeroMeanGaussianModel, level)(vp[3]);
      NEW(NonzeroMeanAr1Model, slope)(pp[4], pp[5], vp[7]);
      Ptr<SemilocalLinearTrendStateModel> trend(new SemilocalLinearTrendStateModel(level, slope));
      if (with_samplers) {
        NEW(ZeroMeanGaussianConjSampler, level_sampler)(level.get(), vp[0], vp[1]);
        if (std::isfinite(vp[2])) level_sampler->set_sigma_upper_limit(vp[2]);
        trend->set_method(level_sampler);
        NEW(GaussianModel, slope_mean_prior)(pp[0], pp[1]);
        NEW(GaussianModel, slope_ar_prior)(pp[2], pp[3]);
        NEW(ChisqModel, slope_sigma_prior)(vp[4], vp[5]);
        NEW(NonzeroMeanAr1Sampler, slope_sampler)(slope.get(), slope_mean_prior, slope_ar_prior,
                                                  slope_sigma_prior);
        if (std::isfinite(vp[6])) slope_sampler->set_sigma_upper_limit(vp[6]);
        if (iparams[3 * b]) slope_sampler->force_stationary();
        if (iparams[3 * b + 1]) slope_sampler->force_ar1_positive();
        trend->set_method(slope_sampler);
      }
      trend->set_initial_level_mean(a0[first]);
      trend->set_initial_slope_mean(a0[first + 1]);
      trend->set_initial_level_sd(std::sqrt(P0[first]));
      trend->set_initial_slope_sd(std::sqrt(P0[first + 1]));
      model->add_state(trend);
      G.slot.push_back((int)G.semilocal.size());
      G.semilocal.push_back(trend);
      G.semilocal_slope.push_back(slope);
      G.semilocal_level.push_back(level);
      first += 3;
    } else if (kinds[b] == 6) {
      const int nf = iparams[3 * b];
      Ptr<TrigStateModel> trig(new TrigStateModel(phi0[16 * b], make_vector(nf, phi0 + 16 * b + 1)));
      trig->error_distribution()->set_sigsq(vp[3] * vp[3]);
      if (with_samplers) {
        NEW(ChisqModel, innovation_precision_prior)(vp[0], vp[1]);
        NEW(ZeroMeanGaussianConjSampler, s)(trig->error_distribution(), innovation_precision_prior);
        if (std::isfinite(vp[2])) s->set_sigma_upper_limit(vp[2]);
        trig->set_method(s);
      }
      const int n = 2 * nf;
      Vector mean(n);
      SpdMatrix var(n, 0.0);
      for (int i = 0; i < n; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
      trig->set_initial_state_mean(mean);
      trig->set_initial_state_variance(var);
      model->add_state(trig);
      G.slot.push_back((int)G.trig.size());
      G.trig.push_back(trig);
      first += n;
    } else {
      const int L = iparams[3 * b];
      Ptr<ArStateModel> arm(new ArStateModel(L));
      arm->set_phi(make_vector(L, phi0 + 16 * b));
      arm->set_sigma(vp[3]);
      if (with_samplers) {
        NEW(ChisqModel, ar_prior)(vp[0], vp[1]);
        NEW(ArPosteriorSampler, s)(arm.get(), ar_prior);
        if (std::isfinite(vp[2])) s->set_sigma_upper_limit(vp[2]);
        arm->set_method(s);
      }
      Vector mean(L);
      SpdMatrix var(L, 0.0);
      for (int i = 0; i < L; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
      arm->set_initial_state_mean(mean);
      arm->set_initial_state_variance(var);
      model->add_state(arm);
      G.slot.push_back((int)G.ar.size());
      G.ar.push_back(arm);
      first += L;
    }
  }
}

int ref_ssg_run(int T, int p, const double *y, const double *X, const uint8_t *observed,
                const double *prior_mean, const double *ominv, double prior_df,
                double sigma_guess, const double *pi, const RefSsvsOptions *opt, int nblocks,
                const int *kinds, const int *iparams, const double *vpar, const double *phi0,
                const double *a0, const double *P0, uint64_t seed, const uint8_t *init_gamma,
                int nsweeps, int state_every, uint8_t *out_gamma, double *out_beta,
                double *out_sigsq, double *out_variances, double *out_phi, double *out_state) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  std::vector<bool> obs;
  if (observed) {
    obs.resize(T);
    for (int t = 0; t < T; ++t) obs[t] = observed[t] != 0;
  }
  NEW(StateSpaceRegressionModel, model)(make_vector(T, y), make_matrix(T, p, X), obs);
  RegressionModel *reg = model->observation_model();
  NEW(MvnGivenScalarSigma, slab)(make_vector(p, prior_mean), make_spd(p, ominv),
                                 reg->Sigsq_prm());
  NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  NEW(BregVsSampler, reg_sampler)(reg, slab, siginv_prior, spike);
  apply_options(*reg_sampler, spike, opt);
  reg->set_method(reg_sampler);
  reg->coef().drop_all();
  for (int j = 0; j < p; ++j) {
    if (init_gamma[j]) reg->coef().add(j);
  }
  GeneralState G;
  add_general_state(model.get(), G, nblocks, kinds, iparams, vpar, phi0, a0, P0, true);
  NEW(StateSpacePosteriorSampler, sampler)(model.get());
  model->set_method(sampler);
  const int m = model->state_dimension();
  size_t kept = 0;
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    record(*reg, p, i, out_gamma, out_beta, out_sigsq);
    for (int b = 0; b < nblocks; ++b) {
      double *v = out_variances + ((size_t)i * nblocks + b) * 2;
      double *ph = out_phi + ((size_t)i * nblocks + b) * 16;
      v[0] = v[1] = 0.0;
      for (int j = 0; j < 16; ++j) ph[j] = 0.0;
      const int s = G.slot[b];
      if (G.kind[b] == 1) {
        v[0] = G.level[s]->sigsq();
      } else if (G.kind[b] == 2) {
        v[0] = G.llt[s]->Sigma()(0, 0);
        v[1] = G.llt[s]->Sigma()(1, 1);
      } else if (G.kind[b] == 3) {
        v[0] = G.seasonal[s]->sigsq();
      } else if (G.kind[b] == 5) {
        // (no parameter)
      } else if (G.kind[b] == 6) {
        v[0] = G.trig[s]->error_distribution()->sigsq();
      } else if (G.kind[b] == 7) {
        v[0] = G.semilocal_level[s]->sigsq();
        v[1] = G.semilocal_slope[s]->sigsq();
        ph[0] = G.semilocal_slope[s]->phi();
        ph[1] = G.semilocal_slope[s]->mu();
      } else {
        v[0] = G.ar[s]->sigsq();
        const int L = iparams[3 * b];
        for (int j = 0; j < L; ++j) ph[j] = G.ar[s]->phi()[j];
      }
    }
    if (state_every > 0 && i % state_every == state_every - 1) {
      const Matrix &state(model->state());
      for (int t = 0; t < T; ++t)
        for (int j = 0; j < m; ++j) out_state[(kept * T + t) * m + j] = state(j, t);
      ++kept;
    }
  }
  REF_CATCH
}

// simulate_forecast of the general model: fixed parameters (sigsq[2 b + v], phi[16 b ..])
// and final state
int ref_ssg_forecast(int T, int p, const double *y, const double *X, const double *beta,
                     const uint8_t *gamma, double sigsq_obs, int nblocks, const int *kinds,
                     const int *iparams, const double *sigsq, const double *phi,
                     const double *final_state, int horizon, const double *newX, uint64_t seed,
                     double *out) {
  REF_TRY
  NEW(StateSpaceRegressionModel, model)(make_vector(T, y), make_matrix(T, p, X),
                                        std::vector<bool>());
  RegressionModel *reg = model->observation_model();
  reg->coef().drop_all();
  Vector b(p, 0.0);
  for (int j = 0; j < p; ++j) {
    if (gamma[j]) {
      reg->coef().add(j);
      b[j] = beta[j];
    }
  }
  reg->coef().set_Beta(b);
  reg->set_sigsq(sigsq_obs);
  int m = 0;
  std::vector<double> vpar(8 * (size_t)nblocks, 1.0);
  for (int k = 0; k < nblocks; ++k) {
    vpar[8 * k + 3] = std::sqrt(sigsq[2 * k]);
    vpar[8 * k + 7] = std::sqrt(sigsq[2 * k + 1]);
    m += (kinds[k] == 1 || kinds[k] == 5) ? 1 : kinds[k] == 2 ? 2 : kinds[k] == 3 ? iparams[3 * k] - 1
         : kinds[k] == 6 ? 2 * iparams[3 * k] : kinds[k] == 7 ? 3 : iparams[3 * k];
  }
  std::vector<double> a0(m, 0.0), P0(m, 1.0);
  GeneralState G;
  add_general_state(model.get(), G, nblocks, kinds, iparams, vpar.data(), phi, a0.data(),
                    P0.data(), false);
  RNG rng(seed);
  Vector fs(m);
  for (int i = 0; i < m; ++i) fs[i] = final_state[i];
  Vector ans = model->simulate_forecast(rng, make_matrix(horizon, p, newX), fs);
  for (int i = 0; i < horizon; ++i) out[i] = ans[i];
  REF_CATCH
}

// ------------------------------------------- SpikeSlabSampler (sigma given)
// The sigma^2-conditional SSVS helper used by the logit / probit / Poisson /
// Student samplers (SpikeSlabSampler.cpp:40-82, 115-138, 171-216), driven the
// way those samplers drive it: per iteration draw_model_indicators(rng, suf,
// sigsq) then draw_beta(rng, suf, sigsq).  slab_kind 0: MvnModel(mu, ivar)
// (precision independent of sigma^2); 1: MvnGivenScalarSigma(mu, ominv) whose
// siginv() is ominv / sigsq.  The weighted sufficient statistics come from
// (X, y, w).  sigsq[i] is the residual variance handed to iteration i.
int ref_sss_run(int n, int p, const double *X, const double *y, const double *w,
                int slab_kind, const double *mu, const double *prec,
                const double *pi, int64_t max_model_size, int max_flips,
                uint64_t seed, const uint8_t *init_gamma, int nsweeps,
                const double *sigsq, uint8_t *out_gamma, double *out_beta) {
  REF_TRY
  Matrix Xm = make_matrix(n, p, X);
  Vector yv = make_vector(n, y);
  Vector wv = w ? make_vector(n, w) : Vector(n, 1.0);
  // (the (X, y, w) constructor prepends an intercept column; recompute() takes
  // the design matrix as it is)
  WeightedRegSuf suf(p);
  suf.recompute(Xm, yv, wv);
  NEW(RegressionModel, model)(p);
  Ptr<MvnBase> slab;
  if (slab_kind == 0) {
    slab = new MvnModel(make_vector(p, mu), make_spd(p, prec), true);
  } else {
    slab = new MvnGivenScalarSigma(make_vector(p, mu), make_spd(p, prec),
                                   model->Sigsq_prm());
  }
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
  SpikeSlabSampler sam(model.get(), slab, spike);
  if (max_flips >= 0) sam.limit_model_selection(max_flips);
  model->coef().drop_all();
  for (int j = 0; j < p; ++j)
    if (init_gamma[j]) model->coef().add(j);
  RNG rng(seed);
  for (int i = 0; i < nsweeps; ++i) {
    model->set_sigsq(sigsq[i]);
    sam.draw_model_indicators(rng, suf, sigsq[i]);
    sam.draw_beta(rng, suf, sigsq[i]);
    const Selector &inc(model->coef().inc());
    const Vector &beta(model->Beta());
    for (int j = 0; j < p; ++j) {
      out_gamma[(size_t)i * p + j] = inc[j] ? 1 : 0;
      out_beta[(size_t)i * p + j] = beta[j];
    }
  }
  REF_CATCH
}

// BinomialProbitSpikeSlabSampler (probit): data augmentation
// (BinomialProbitDataImputer) + SpikeSlabSampler on the complete-data sufficient
// statistics.  X is n x p column-major, y successes, ntrials trials.
int ref_probit_run(int n, int p, const double *X, const double *y, const double *ntrials,
                   const double *mu, const double *prec, const double *pi,
                   int64_t max_model_size, int max_flips, int clt_threshold, uint64_t seed,
                   const uint8_t *init_gamma, const double *init_beta, int nsweeps,
                   uint8_t *out_gamma, double *out_beta) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  NEW(BinomialProbitModel, model)(p);
  for (int i = 0; i < n; ++i) {
    Vector x(p);
    for (int j = 0; j < p; ++j) x[j] = X[(size_t)j * n + i];
    NEW(BinomialRegressionData, dp)(y[i], ntrials[i], x);
    model->add_data(dp);
  }
  Ptr<MvnBase> slab(new MvnModel(make_vector(p, mu), make_spd(p, prec), true));
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
  NEW(BinomialProbitSpikeSlabSampler, sam)(model.get(), slab, spike, clt_threshold);
  if (max_flips >= 0) sam->limit_model_selection(max_flips);
  model->set_method(sam);
  model->coef().drop_all();
  Vector b0(p, 0.0);
  for (int j = 0; j < p; ++j)
    if (init_gamma[j]) {
      model->coef().add(j);
      b0[j] = init_beta[j];
    }
  model->coef().set_Beta(b0);
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    const Selector &inc(model->coef().inc());
    const Vector &beta(model->Beta());
    for (int j = 0; j < p; ++j) {
      out_gamma[(size_t)i * p + j] = inc[j] ? 1 : 0;
      out_beta[(size_t)i * p + j] = beta[j];
    }
  }
  REF_CATCH
}

// BinomialLogitSpikeSlabSampler (logit)
int ref_logit_run(int n, int p, const double *X, const double *y, const double *ntrials,
                  const double *mu, const double *prec, const double *pi,
                  int64_t max_model_size, int max_flips, int clt_threshold, uint64_t seed,
                  const uint8_t *init_gamma, const double *init_beta, int nsweeps,
                  uint8_t *out_gamma, double *out_beta) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  NEW(BinomialLogitModel, model)(p);
  for (int i = 0; i < n; ++i) {
    Vector x(p);
    for (int j = 0; j < p; ++j) x[j] = X[(size_t)j * n + i];
    NEW(BinomialRegressionData, dp)(y[i], ntrials[i], x);
    model->add_data(dp);
  }
  Ptr<MvnBase> slab(new MvnModel(make_vector(p, mu), make_spd(p, prec), true));
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
  NEW(BinomialLogitSpikeSlabSampler, sam)(model.get(), slab, spike, clt_threshold);
  if (max_flips >= 0) sam->limit_model_selection(max_flips);
  model->set_method(sam);
  model->coef().drop_all();
  Vector b0(p, 0.0);
  for (int j = 0; j < p; ++j)
    if (init_gamma[j]) {
      model->coef().add(j);
      b0[j] = init_beta[j];
    }
  model->coef().set_Beta(b0);
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    const Selector &inc(model->coef().inc());
    const Vector &beta(model->Beta());
    for (int j = 0; j < p; ++j) {
      out_gamma[(size_t)i * p + j] = inc[j] ? 1 : 0;
      out_beta[(size_t)i * p + j] = beta[j];
    }
  }
  REF_CATCH
}

// ---- PoissonRegressionSpikeSlabSampler (the Poisson member) --------
// The normal-mixture approximations of NegLogGamma(n) the reference's table yields
// (create_poisson_mixture_approximation_table + approximate(n)), DATA for the oracle, the
// device and the golden fixtures.  approximate(n) interpolates between the tabulated
// orders and, where that is too coarse, refits and ADDS the result to the table, so that
// what a later request returns can depend on the earlier ones: `requests` replays the
// order in which a sampler asks (for every observation: 1, then its count if positive);
// `counts` (ascending, distinct) are then read from the table.  Out arrays hold up to
// max_comp components per count; a count at or beyond the table's largest order has
// none (the Gaussian limit is used).
int ref_poisson_mixtures(int nrequests, const int64_t *requests, int ncounts, const int64_t *counts,
                         int max_comp, int *ncomp, double *mu, double *sigma, double *weight,
                         int64_t *largest_index) {
  REF_TRY
  NormalMixtureApproximationTable table = create_poisson_mixture_approximation_table();
  *largest_index = table.largest_index();
  for (int i = 0; i < nrequests; ++i)
    if (requests[i] > 0 && requests[i] < table.largest_index()) table.approximate((int)requests[i]);
  for (int i = 0; i < ncounts; ++i) {
    if (counts[i] >= table.largest_index()) {
      ncomp[i] = 0;
      continue;
    }
    const NormalMixtureApproximation &a(table.approximate((int)counts[i]));
    if (a.dim() > max_comp) throw std::runtime_error("more mixture components than room");
    ncomp[i] = a.dim();
    for (int c = 0; c < a.dim(); ++c) {
      mu[(size_t)i * max_comp + c] = a.mu()[c];
      sigma[(size_t)i * max_comp + c] = a.sigma()[c];
      weight[(size_t)i * max_comp + c] = a.weights()[c];
    }
  }
  REF_CATCH
}

int ref_poisson_run(int n, int p, const double *X, const double *y, const double *exposure,
                    const double *mu, const double *prec, const double *pi,
                    int64_t max_model_size, int max_flips, uint64_t seed,
                    const uint8_t *init_gamma, const double *init_beta, int nsweeps,
                    uint8_t *out_gamma, double *out_beta) {
  REF_TRY
  GlobalRng::rng.seed(seed);
  NEW(PoissonRegressionModel, model)(p);
  for (int i = 0; i < n; ++i) {
    Vector x(p);
    for (int j = 0; j < p; ++j) x[j] = X[(size_t)j * n + i];
    NEW(PoissonRegressionData, dp)((int64_t)llround(y[i]), x, exposure[i]);
    model->add_data(dp);
  }
  Ptr<MvnBase> slab(new MvnModel(make_vector(p, mu), make_spd(p, prec), true));
  NEW(VariableSelectionPrior, spike)(make_vector(p, pi));
  if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
  NEW(PoissonRegressionSpikeSlabSampler, sam)(model.get(), slab, spike, 1);
  if (max_flips >= 0) sam->limit_model_selection(max_flips);
  model->set_method(sam);
  model->coef().drop_all();
  Vector b0(p, 0.0);
  for (int j = 0; j < p; ++j)
    if (init_gamma[j]) {
      model->coef().add(j);
      b0[j] = init_beta[j];
    }
  model->coef().set_Beta(b0);
  for (int i = 0; i < nsweeps; ++i) {
    model->sample_posterior();
    const Selector &inc(model->coef().inc());
    const Vector &beta(model->Beta());
    for (int j = 0; j < p; ++j) {
      out_gamma[(size_t)i * p + j] = inc[j] ? 1 : 0;
      out_beta[(size_t)i * p + j] = beta[j];
    }
  }
  REF_CATCH
}

// two-sided truncated normal draws: rtrun_norm_2_mt(rng, mu, sigma, lo, hi)
int ref_rng_trun_norm_2(uint64_t seed, double mu, double sigma, double lo, double hi, int n,
                        double *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = rtrun_norm_2_mt(rng, mu, sigma, lo, hi);
  REF_CATCH
}

// truncated normal draws: rtrun_norm_mt(rng, mu, sigma, cut, above)
int ref_rng_trun_norm(uint64_t seed, double mu, double sigma, double cut, int above, int n,
                      double *out) {
  REF_TRY
  RNG rng(seed);
  for (int i = 0; i < n; ++i) out[i] = rtrun_norm_mt(rng, mu, sigma, cut, above != 0);
  REF_CATCH
}

// weighted sufficient statistics (WeightedRegSuf(X, y, w))
int ref_weighted_suf(int n, int p, const double *X, const double *y,
                     const double *w, double *xtx, double *xty) {
  REF_TRY
  WeightedRegSuf suf(p);
  suf.recompute(make_matrix(n, p, X), make_vector(n, y),
                w ? make_vector(n, w) : Vector(n, 1.0));
  SpdMatrix S = suf.xtx();
  std::memcpy(xtx, S.data(), sizeof(double) * p * p);
  Vector s = suf.xty();
  std::memcpy(xty, s.data(), sizeof(double) * p);
  REF_CATCH
}

}  // extern "C"
