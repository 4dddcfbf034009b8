// Test driver for the reference-side binding: the REFERENCE's RegressionModel,
// priors and `model->sample_posterior()` loop with DeviceBregVsSampler as the
// sampling method -- the drop-in in the reference's own words.  Exported with C
// linkage so that a -m gpu test can run it through ctypes on the GPU box and
// compare chain 0 (what the BOOM model object sees) with the oracle.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "DeviceBinomialLogitSpikeSlabSampler.hpp"
#include "DeviceBregVsSampler.hpp"
#include "DevicePoissonRegressionSpikeSlabSampler.hpp"
#include "DeviceStateSpacePosteriorSampler.hpp"
#include "Models/StateSpace/StateModels/LocalLevelStateModel.hpp"
#include "Models/StateSpace/StateModels/LocalLinearTrend.hpp"
#include "Models/StateSpace/StateModels/SeasonalStateModel.hpp"
#include "Models/StateSpace/StateModels/StaticInterceptStateModel.hpp"
#include "Models/StateSpace/StateModels/SemilocalLinearTrend.hpp"
#include "Models/TimeSeries/NonzeroMeanAr1Model.hpp"
#include "Models/GaussianModel.hpp"
#include "Models/StateSpace/StateModels/TrigStateModel.hpp"
#include "Models/MvnModel.hpp"
#include "LinAlg/Matrix.hpp"
#include "LinAlg/SpdMatrix.hpp"
#include "Models/ChisqModel.hpp"
#include "distributions/rng.hpp"

using namespace BOOM;

static std::string g_binding_error;
// (the next ref_binding_ss_run: before draw `change_at` the spike's inclusion probabilities and
// the slab's mean are set to these, through the prior objects' own setters)
static int g_ss_change_at = -1;
static Vector g_ss_pi2, g_ss_mu2;

extern "C" {

const char *ref_binding_last_error() { return g_binding_error.c_str(); }

void ref_binding_ss_change_priors(int change_at, int p, const double *pi2, const double *mu2) {
  g_ss_change_at = change_at;
  g_ss_pi2 = Vector(p);
  g_ss_mu2 = Vector(p);
  for (int j = 0; j < p; ++j) { g_ss_pi2[j] = pi2[j]; g_ss_mu2[j] = mu2[j]; }
}

// X n x p column-major.  Returns 0 or -1 (message in ref_binding_last_error).
// out_seed receives the seed the binding gave the engine (seed_rng(GlobalRng)
// after GlobalRng::rng.seed(seed)), i.e. the key of the oracle run to compare with.
// ndevices == 0: the single-device constructor on device 0; ndevices >= 1: the
// device-list constructor with device 0 repeated ndevices times and `chains` chains
// per entry (a one-GPU box can check the list path: the draws of a global chain do not
// depend on the list).
static int binding_run_impl(int n, int p, const double *X, const double *y,
                    const double *prior_mean, const double *ominv, double prior_df,
                    double sigma_guess, const double *pi, int64_t max_model_size,
                    double sigma_upper_limit, int max_flips, double swap_threshold,
                    int chains, int ndevices, int lookahead, uint64_t seed,
                    const uint8_t *init_gamma,
                    int nsweeps, uint8_t *out_gamma, double *out_beta, double *out_sigsq,
                    double *out_logpri, uint64_t *out_seed,
                    int probe_chain, uint8_t *probe_gamma, double *probe_beta,
                    double *probe_sigsq) {
  try {
    GlobalRng::rng.seed(seed);
    Matrix Xm(n, p);
    for (int j = 0; j < p; ++j)
      for (int i = 0; i < n; ++i) Xm(i, j) = X[(size_t)j * n + i];
    Vector yv(n);
    for (int i = 0; i < n; ++i) yv[i] = y[i];
    Ptr<RegressionModel> model(new RegressionModel(Xm, yv, false));
    Vector mu(p);
    SpdMatrix om(p);
    Vector piv(p);
    for (int j = 0; j < p; ++j) {
      mu[j] = prior_mean[j];
      piv[j] = pi[j];
      for (int i = 0; i < p; ++i) om(i, j) = ominv[(size_t)j * p + i];
    }
    NEW(MvnGivenScalarSigma, slab)(mu, om, model->Sigsq_prm());
    NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
    NEW(VariableSelectionPrior, spike)(piv);
    if (max_model_size >= 0) spike->set_max_model_size(max_model_size);
    model->coef().drop_all();
    for (int j = 0; j < p; ++j)
      if (init_gamma[j]) model->coef().add(j);
    Ptr<DeviceBregVsSampler> sampler;
    if (ndevices <= 0) {
      sampler.reset(new DeviceBregVsSampler(model.get(), slab, siginv_prior, spike, chains,
                                            0, lookahead));
    } else {
      sampler.reset(new DeviceBregVsSampler(model.get(), slab, siginv_prior, spike, chains,
                                            std::vector<int>(ndevices, 0), lookahead));
    }
    if (std::isfinite(sigma_upper_limit)) sampler->set_sigma_upper_limit(sigma_upper_limit);
    if (max_flips >= 0) sampler->limit_model_selection(max_flips);
    if (swap_threshold != 0.8) sampler->set_correlation_swap_threshold(swap_threshold);
    if (out_seed) *out_seed = sampler->device_seed();
    model->set_method(sampler);
    for (int s = 0; s < nsweeps; ++s) {
      model->sample_posterior();   // PriorPolicy::sample_posterior -> sampler->draw()
      const Selector &inc(model->coef().inc());
      const Vector beta = model->Beta();
      for (int j = 0; j < p; ++j) {
        out_gamma[(size_t)s * p + j] = inc[j] ? 1 : 0;
        out_beta[(size_t)s * p + j] = beta[j];
      }
      out_sigsq[s] = model->sigsq();
      if (out_logpri) out_logpri[s] = sampler->logpri();
    }
    if (probe_gamma) {
      Selector inc(p, false);
      Vector beta;
      double s2 = 0;
      sampler->chain_state(probe_chain, inc, beta, s2);
      for (int j = 0; j < p; ++j) {
        probe_gamma[j] = inc[j] ? 1 : 0;
        probe_beta[j] = beta[j];
      }
      *probe_sigsq = s2;
    }
    return 0;
  } catch (std::exception &e) {
    g_binding_error = e.what();
    return -1;
  }
}

int ref_binding_run(int n, int p, const double *X, const double *y,
                    const double *prior_mean, const double *ominv, double prior_df,
                    double sigma_guess, const double *pi, int64_t max_model_size,
                    double sigma_upper_limit, int max_flips, double swap_threshold,
                    int chains, int lookahead, uint64_t seed, const uint8_t *init_gamma,
                    int nsweeps, uint8_t *out_gamma, double *out_beta, double *out_sigsq,
                    double *out_logpri, uint64_t *out_seed,
                    int probe_chain, uint8_t *probe_gamma, double *probe_beta,
                    double *probe_sigsq) {
  return binding_run_impl(n, p, X, y, prior_mean, ominv, prior_df, sigma_guess, pi,
                          max_model_size, sigma_upper_limit, max_flips, swap_threshold,
                          chains, 0, lookahead, seed, init_gamma, nsweeps, out_gamma,
                          out_beta, out_sigsq, out_logpri, out_seed, probe_chain,
                          probe_gamma, probe_beta, probe_sigsq);
}

// The device-list constructor (see binding_run_impl); probe_chain is a GLOBAL chain id
int ref_binding_group_run(int n, int p, const double *X, const double *y,
                    const double *prior_mean, const double *ominv, double prior_df,
                    double sigma_guess, const double *pi, int64_t max_model_size,
                    double sigma_upper_limit, int max_flips, double swap_threshold,
                    int chains_per_device, int ndevices, int lookahead, uint64_t seed,
                    const uint8_t *init_gamma,
                    int nsweeps, uint8_t *out_gamma, double *out_beta, double *out_sigsq,
                    double *out_logpri, uint64_t *out_seed,
                    int probe_chain, uint8_t *probe_gamma, double *probe_beta,
                    double *probe_sigsq) {
  return binding_run_impl(n, p, X, y, prior_mean, ominv, prior_df, sigma_guess, pi,
                          max_model_size, sigma_upper_limit, max_flips, swap_threshold,
                          chains_per_device, ndevices, lookahead, seed, init_gamma, nsweeps,
                          out_gamma, out_beta, out_sigsq, out_logpri, out_seed, probe_chain,
                          probe_gamma, probe_beta, probe_sigsq);
}

// Priors changed under the sampler (ctor #5's purpose, BregVsSampler.hpp:98-101): the
// caller keeps the Ptrs; after `change_at` draws it sets new prior inclusion probabilities
// on the spike, a new mean on the slab and a new sum of squares on the residual prior --
// through the objects' own setters, nothing is said to the sampler -- and goes on drawing.
// ndevices as in binding_run_impl.
int ref_binding_mutating_priors_run(int n, int p, const double *X, const double *y,
                                    const double *prior_mean, const double *ominv, double prior_df,
                                    double sigma_guess, const double *pi, const double *prior_mean2,
                                    double sigma_guess2, const double *pi2, int change_at, int chains,
                                    int ndevices, int lookahead, uint64_t seed,
                                    const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                                    double *out_beta, double *out_sigsq, uint64_t *out_seed) {
  try {
    GlobalRng::rng.seed(seed);
    Matrix Xm(n, p);
    for (int j = 0; j < p; ++j)
      for (int i = 0; i < n; ++i) Xm(i, j) = X[(size_t)j * n + i];
    Vector yv(n);
    for (int i = 0; i < n; ++i) yv[i] = y[i];
    Ptr<RegressionModel> model(new RegressionModel(Xm, yv, false));
    Vector mu(p), mu2(p), piv(p), piv2(p);
    SpdMatrix om(p);
    for (int j = 0; j < p; ++j) {
      mu[j] = prior_mean[j];
      mu2[j] = prior_mean2[j];
      piv[j] = pi[j];
      piv2[j] = pi2[j];
      for (int i = 0; i < p; ++i) om(i, j) = ominv[(size_t)j * p + i];
    }
    NEW(MvnGivenScalarSigma, slab)(mu, om, model->Sigsq_prm());
    NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
    NEW(VariableSelectionPrior, spike)(piv);
    model->coef().drop_all();
    for (int j = 0; j < p; ++j)
      if (init_gamma[j]) model->coef().add(j);
    Ptr<DeviceBregVsSampler> sampler;
    if (ndevices <= 0) {
      sampler.reset(new DeviceBregVsSampler(model.get(), slab, siginv_prior, spike, chains, 0, lookahead));
    } else {
      sampler.reset(new DeviceBregVsSampler(model.get(), slab, siginv_prior, spike, chains,
                                            std::vector<int>(ndevices, 0), lookahead));
    }
    if (out_seed) *out_seed = sampler->device_seed();
    model->set_method(sampler);
    for (int s = 0; s < nsweeps; ++s) {
      if (s == change_at) {
        spike->set_prior_inclusion_probabilities(piv2);
        slab->set_mu(mu2);
        siginv_prior->set_sigma_estimate(sigma_guess2);   // ChisqModel: the same df, another guess
      }
      model->sample_posterior();
      const Selector &inc(model->coef().inc());
      const Vector beta = model->Beta();
      for (int j = 0; j < p; ++j) {
        out_gamma[(size_t)s * p + j] = inc[j] ? 1 : 0;
        out_beta[(size_t)s * p + j] = beta[j];
      }
      out_sigsq[s] = model->sigsq();
    }
    return 0;
  } catch (std::exception &e) {
    g_binding_error = e.what();
    return -1;
  }
}

// BregVsSampler's constructors #1 - #4 on the device sampler (which: 1 .. 4).  #1: a =
// {prior_nobs, expected_rsq, expected_model_size}, flag = first_term_is_intercept; #2: a =
// {prior_sigma_nobs, prior_sigma_guess, prior_beta_nobs, diagonal_shrinkage,
// prior_inclusion_probability}, flag = force_intercept; #3 / #4: the numbers of
// ref_binding_run (as arguments / as a ZellnerPriorParameters).
int ref_binding_ctor_run(int which, int n, int p, const double *X, const double *y, const double *a, int flag,
                         const double *prior_mean, const double *ominv, double prior_df, double sigma_guess,
                         const double *pi, int chains, int lookahead, uint64_t seed,
                         const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma, double *out_beta,
                         double *out_sigsq, uint64_t *out_seed) {
  try {
    GlobalRng::rng.seed(seed);
    Matrix Xm(n, p);
    for (int j = 0; j < p; ++j)
      for (int i = 0; i < n; ++i) Xm(i, j) = X[(size_t)j * n + i];
    Vector yv(n);
    for (int i = 0; i < n; ++i) yv[i] = y[i];
    Ptr<RegressionModel> model(new RegressionModel(Xm, yv, false));
    model->coef().drop_all();
    for (int j = 0; j < p; ++j)
      if (init_gamma[j]) model->coef().add(j);
    Ptr<DeviceBregVsSampler> sampler;
    if (which == 1) {
      sampler.reset(new DeviceBregVsSampler(model.get(), a[0], a[1], a[2], flag != 0, chains, 0, lookahead));
    } else if (which == 2) {
      sampler.reset(new DeviceBregVsSampler(model.get(), a[0], a[1], a[2], a[3], a[4], flag != 0, chains, 0,
                                            lookahead));
    } else {
      Vector mu(p), piv(p);
      SpdMatrix om(p);
      for (int j = 0; j < p; ++j) {
        mu[j] = prior_mean[j];
        piv[j] = pi[j];
        for (int i = 0; i < p; ++i) om(i, j) = ominv[(size_t)j * p + i];
      }
      if (which == 3) {
        sampler.reset(new DeviceBregVsSampler(model.get(), mu, om, sigma_guess, prior_df, piv, chains, 0,
                                              lookahead));
      } else {
        ZellnerPriorParameters z;
        z.prior_inclusion_probabilities = piv;
        z.prior_beta_guess = mu;
        z.prior_beta_information = om;
        z.prior_sigma_guess = sigma_guess;
        z.prior_sigma_guess_weight = prior_df;
        sampler.reset(new DeviceBregVsSampler(model.get(), z, chains, 0, lookahead));
      }
    }
    if (out_seed) *out_seed = sampler->device_seed();
    model->set_method(sampler);
    for (int s = 0; s < nsweeps; ++s) {
      model->sample_posterior();
      const Selector &inc(model->coef().inc());
      const Vector beta = model->Beta();
      for (int j = 0; j < p; ++j) {
        out_gamma[(size_t)s * p + j] = inc[j] ? 1 : 0;
        out_beta[(size_t)s * p + j] = beta[j];
      }
      out_sigsq[s] = model->sigsq();
    }
    return 0;
  } catch (std::exception &e) {
    g_binding_error = e.what();
    return -1;
  }
}

// The same for the logit sampler: BOOM's BinomialLogitModel (data added one
// BinomialRegressionData at a time, as the reference's callers do), MvnModel slab,
// VariableSelectionPrior, and sample_posterior() with the device sampler attached.
int ref_binding_logit_run(int n, int p, const double *X, const double *y, const double *ntrials,
                          const double *slab_mean, const double *slab_precision, const double *pi,
                          int clt_threshold, int max_flips, int chains, uint64_t seed,
                          const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                          double *out_beta, uint64_t *out_seed, int probe_chain,
                          uint8_t *probe_gamma, double *probe_beta, int ndevices) {
  try {
    GlobalRng::rng.seed(seed);
    Ptr<BinomialLogitModel> model(new BinomialLogitModel(p, true));
    for (int i = 0; i < n; ++i) {
      Vector x(p);
      for (int j = 0; j < p; ++j) x[j] = X[(size_t)j * n + i];
      NEW(BinomialRegressionData, dp)(y[i], ntrials[i], x);
      model->add_data(dp);
    }
    Vector mu(p), piv(p);
    SpdMatrix prec(p);
    for (int j = 0; j < p; ++j) {
      mu[j] = slab_mean[j];
      piv[j] = pi[j];
      for (int i = 0; i < p; ++i) prec(i, j) = slab_precision[(size_t)j * p + i];
    }
    NEW(MvnModel, slab)(mu, prec, true);
    NEW(VariableSelectionPrior, spike)(piv);
    model->coef().drop_all();
    for (int j = 0; j < p; ++j)
      if (init_gamma[j]) model->coef().add(j);
    // (ndevices > 0: `chains` per entry of a device list that names device 0 ndevices times)
    Ptr<DeviceBinomialLogitSpikeSlabSampler> sampler(
        ndevices > 0 ? new DeviceBinomialLogitSpikeSlabSampler(model.get(), slab, spike, clt_threshold, chains,
                                                               std::vector<int>(ndevices, 0))
                     : new DeviceBinomialLogitSpikeSlabSampler(model.get(), slab, spike, clt_threshold, chains));
    if (max_flips > 0) sampler->limit_model_selection(max_flips);
    if (out_seed) *out_seed = sampler->device_seed();
    model->set_method(sampler);
    for (int s = 0; s < nsweeps; ++s) {
      model->sample_posterior();
      const Selector &inc(model->coef().inc());
      const Vector beta = model->Beta();
      for (int j = 0; j < p; ++j) {
        out_gamma[(size_t)s * p + j] = inc[j] ? 1 : 0;
        out_beta[(size_t)s * p + j] = beta[j];
      }
    }
    if (probe_gamma) {
      Selector inc(p, false);
      Vector beta;
      sampler->chain_state(probe_chain, inc, beta);
      for (int j = 0; j < p; ++j) {
        probe_gamma[j] = inc[j] ? 1 : 0;
        probe_beta[j] = beta[j];
      }
    }
    return 0;
  } catch (std::exception &e) {
    g_binding_error = e.what();
    return -1;
  }
}

// The bsts half: BOOM's StateSpaceRegressionModel(y, X, observed) with a
// LocalLevelStateModel (trend = 1) or LocalLinearTrendStateModel (trend = 2) and an
// optional SeasonalStateModel(nseasons) added with add_state, stepped by
// model->sample_posterior() with DeviceStateSpacePosteriorSampler as the model's
// sampling method.  Recorded after every draw, from the BOOM objects: the regression
// model's inc / Beta / sigsq, the state models' variances (level, slope, seasonal;
// unused 0) and model->state() (T x m per draw, step t at [t m, (t + 1) m)).
// Three-element arrays are indexed level, slope, seasonal.
// ar_lags > 0: an ArStateModel(ar_lags) comes last; ar[] = {prior df, prior sigma guess,
// sigma upper limit, initial sigma}, ar_phi0 its initial coefficients, its initial state
// moments follow the others; out_ar: nsweeps x (ar_lags + 1) = phi, sigsq as the BOOM
// ArStateModel object holds them after every draw
static int binding_ss_impl(int T, int p, const double *y, const double *X, const uint8_t *observed,
                       const double *prior_mean, const double *ominv, double prior_df,
                       double sigma_guess, const double *pi, double sigma_upper_limit,
                       int trend, int nseasons, const double *var_df,
                       const double *var_sigma_guess, const double *var_sigma_upper_limit,
                       const double *var_initial_sigma, const double *initial_state_mean,
                       const double *initial_state_variance, int ar_lags, const double *ar,
                       const double *ar_phi0, int chains, uint64_t seed,
                       const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                       double *out_beta, double *out_sigsq, double *out_variances,
                       double *out_state, double *out_ar, double *out_logpri, uint64_t *out_seed,
                       int probe_chain, uint8_t *probe_gamma, double *probe_state) {
  try {
    GlobalRng::rng.seed(seed);
    Matrix Xm(T, p);
    for (int j = 0; j < p; ++j)
      for (int t = 0; t < T; ++t) Xm(t, j) = X[(size_t)j * T + t];
    Vector yv(T);
    for (int t = 0; t < T; ++t) yv[t] = y[t];
    std::vector<bool> obs;
    if (observed) {
      obs.resize(T);
      for (int t = 0; t < T; ++t) obs[t] = observed[t] != 0;
    }
    NEW(StateSpaceRegressionModel, model)(yv, Xm, obs);
    RegressionModel *reg = model->observation_model();
    Vector mu(p), piv(p);
    SpdMatrix om(p);
    for (int j = 0; j < p; ++j) {
      mu[j] = prior_mean[j];
      piv[j] = pi[j];
      for (int i = 0; i < p; ++i) om(i, j) = ominv[(size_t)j * p + i];
    }
    NEW(MvnGivenScalarSigma, slab)(mu, om, reg->Sigsq_prm());
    NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
    NEW(VariableSelectionPrior, spike)(piv);
    reg->coef().drop_all();
    for (int j = 0; j < p; ++j)
      if (init_gamma[j]) reg->coef().add(j);

    const int ns1 = nseasons > 0 ? nseasons - 1 : 0, m = trend + ns1 + ar_lags;
    std::vector<DeviceStateVariancePrior> vpriors;
    auto vprior = [&](int slot) {
      DeviceStateVariancePrior pr;
      pr.precision_prior = new ChisqModel(var_df[slot], var_sigma_guess[slot]);
      pr.sigma_upper_limit = var_sigma_upper_limit[slot];
      vpriors.push_back(pr);
    };
    Ptr<LocalLevelStateModel> level;
    Ptr<LocalLinearTrendStateModel> llt;
    Ptr<SeasonalStateModel> seasonal;
    if (trend == 1) {
      level = new LocalLevelStateModel(var_initial_sigma[0]);
      level->set_initial_state_mean(initial_state_mean[0]);
      level->set_initial_state_variance(initial_state_variance[0]);
      model->add_state(level);
      vprior(0);
    } else {
      llt = new LocalLinearTrendStateModel;
      SpdMatrix Sigma(2, 0.0);
      Sigma(0, 0) = var_initial_sigma[0] * var_initial_sigma[0];
      Sigma(1, 1) = var_initial_sigma[1] * var_initial_sigma[1];
      llt->set_Sigma(Sigma);
      Vector a0(2);
      SpdMatrix P0(2, 0.0);
      for (int i = 0; i < 2; ++i) {
        a0[i] = initial_state_mean[i];
        P0(i, i) = initial_state_variance[i];
      }
      llt->set_initial_state_mean(a0);
      llt->set_initial_state_variance(P0);
      model->add_state(llt);
      vprior(0);
      vprior(1);
    }
    if (nseasons > 0) {
      seasonal = new SeasonalStateModel(nseasons, 1);
      seasonal->set_sigsq(var_initial_sigma[2] * var_initial_sigma[2]);
      Vector a0(ns1);
      SpdMatrix P0(ns1, 0.0);
      for (int i = 0; i < ns1; ++i) {
        a0[i] = initial_state_mean[trend + i];
        P0(i, i) = initial_state_variance[trend + i];
      }
      seasonal->set_initial_state_mean(a0);
      seasonal->set_initial_state_variance(P0);
      model->add_state(seasonal);
      vprior(2);
    }
    Ptr<ArStateModel> arm;
    if (ar_lags > 0) {
      arm = new ArStateModel(ar_lags);
      Vector phi0(ar_lags);
      for (int i = 0; i < ar_lags; ++i) phi0[i] = ar_phi0[i];
      arm->set_phi(phi0);
      arm->set_sigma(ar[3]);
      Vector a0(ar_lags);
      SpdMatrix P0(ar_lags, 0.0);
      for (int i = 0; i < ar_lags; ++i) {
        a0[i] = initial_state_mean[trend + ns1 + i];
        P0(i, i) = initial_state_variance[trend + ns1 + i];
      }
      arm->set_initial_state_mean(a0);
      arm->set_initial_state_variance(P0);
      model->add_state(arm);
      DeviceStateVariancePrior pr;
      pr.precision_prior = new ChisqModel(ar[0], ar[1]);
      pr.sigma_upper_limit = ar[2];
      vpriors.push_back(pr);
    }

    NEW(DeviceStateSpacePosteriorSampler, sampler)(model.get(), slab, siginv_prior, spike,
                                                   sigma_upper_limit, vpriors, chains);
    if (out_seed) *out_seed = sampler->device_seed();
    model->set_method(sampler);
    for (int s = 0; s < nsweeps; ++s) {
      if (s == g_ss_change_at && (int)g_ss_pi2.size() == p) {
        // (ref_binding_ss_change_priors: the caller's spike and slab change under the sampler)
        spike->set_prior_inclusion_probabilities(g_ss_pi2);
        slab->set_mu(g_ss_mu2);
        g_ss_change_at = -1;
      }
      model->sample_posterior();   // PriorPolicy::sample_posterior -> sampler->draw()
      const Selector &inc(reg->coef().inc());
      const Vector beta = reg->Beta();
      for (int j = 0; j < p; ++j) {
        out_gamma[(size_t)s * p + j] = inc[j] ? 1 : 0;
        out_beta[(size_t)s * p + j] = beta[j];
      }
      out_sigsq[s] = reg->sigsq();
      double *v = out_variances + (size_t)s * 3;
      v[0] = v[1] = v[2] = 0.0;
      if (level) {
        v[0] = level->sigsq();
      } else {
        v[0] = llt->Sigma()(0, 0);
        v[1] = llt->Sigma()(1, 1);
      }
      if (seasonal) v[2] = seasonal->sigsq();
      if (arm) {
        for (int i = 0; i < ar_lags; ++i) out_ar[(size_t)s * (ar_lags + 1) + i] = arm->phi()[i];
        out_ar[(size_t)s * (ar_lags + 1) + ar_lags] = arm->sigsq();
      }
      const Matrix &state(model->state());
      if (state.nrow() != m || state.ncol() != T) throw std::runtime_error("state has the wrong shape");
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < m; ++i) out_state[((size_t)s * T + t) * m + i] = state(i, t);
      if (out_logpri) out_logpri[s] = sampler->logpri();
    }
    if (probe_gamma) {
      Selector inc(p, false);
      Vector beta, variances;
      Matrix state;
      double s2 = 0;
      sampler->chain_state(probe_chain, inc, beta, s2, variances, state);
      for (int j = 0; j < p; ++j) probe_gamma[j] = inc[j] ? 1 : 0;
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < m; ++i) probe_state[(size_t)t * m + i] = state(i, t);
    }
    return 0;
  } catch (std::exception &e) {
    g_binding_error = e.what();
    return -1;
  }
}

int ref_binding_ss_run(int T, int p, const double *y, const double *X, const uint8_t *observed,
                       const double *prior_mean, const double *ominv, double prior_df,
                       double sigma_guess, const double *pi, double sigma_upper_limit,
                       int trend, int nseasons, const double *var_df,
                       const double *var_sigma_guess, const double *var_sigma_upper_limit,
                       const double *var_initial_sigma, const double *initial_state_mean,
                       const double *initial_state_variance, int chains, uint64_t seed,
                       const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                       double *out_beta, double *out_sigsq, double *out_variances,
                       double *out_state, double *out_logpri, uint64_t *out_seed,
                       int probe_chain, uint8_t *probe_gamma, double *probe_state) {
  return binding_ss_impl(T, p, y, X, observed, prior_mean, ominv, prior_df, sigma_guess, pi,
                         sigma_upper_limit, trend, nseasons, var_df, var_sigma_guess,
                         var_sigma_upper_limit, var_initial_sigma, initial_state_mean,
                         initial_state_variance, 0, nullptr, nullptr, chains, seed, init_gamma,
                         nsweeps, out_gamma, out_beta, out_sigsq, out_variances, out_state, nullptr,
                         out_logpri, out_seed, probe_chain, probe_gamma, probe_state);
}

// the same with an ArStateModel added last (see binding_ss_impl)
int ref_binding_ss_ar_run(int T, int p, const double *y, const double *X, const uint8_t *observed,
                          const double *prior_mean, const double *ominv, double prior_df,
                          double sigma_guess, const double *pi, double sigma_upper_limit,
                          int trend, int nseasons, const double *var_df,
                          const double *var_sigma_guess, const double *var_sigma_upper_limit,
                          const double *var_initial_sigma, const double *initial_state_mean,
                          const double *initial_state_variance, int ar_lags, const double *ar,
                          const double *ar_phi0, int chains, uint64_t seed,
                          const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                          double *out_beta, double *out_sigsq, double *out_variances,
                          double *out_state, double *out_ar, double *out_logpri, uint64_t *out_seed,
                          int probe_chain, uint8_t *probe_gamma, double *probe_state) {
  return binding_ss_impl(T, p, y, X, observed, prior_mean, ominv, prior_df, sigma_guess, pi,
                         sigma_upper_limit, trend, nseasons, var_df, var_sigma_guess,
                         var_sigma_upper_limit, var_initial_sigma, initial_state_mean,
                         initial_state_variance, ar_lags, ar, ar_phi0, chains, seed, init_gamma,
                         nsweeps, out_gamma, out_beta, out_sigsq, out_variances, out_state, out_ar,
                         out_logpri, out_seed, probe_chain, probe_gamma, probe_state);
}

// The general form: ANY list of state models (the flat arrays of oracle/ref_driver.cpp's
// ref_ssg_run: kinds, iparams = {nseasons, duration, time of first observation} or {lags},
// vpar[8 b + 4 v + {df, sigma guess, upper limit, initial sigma}], phi0[16 b ..], a0 / P0).
// Per draw: the model's inc / Beta / sigsq, every state model's variance parameters
// (2 per block) and coefficients (16 per block) as BOOM's own objects hold them after
// sample_posterior(), model->state(), the sampler's logpri(); at the end one other chain.
int ref_binding_ssg_run(int T, int p, const double *y, const double *X, const uint8_t *observed,
                        const double *prior_mean, const double *ominv, double prior_df,
                        double sigma_guess, const double *pi, double sigma_upper_limit, int nblocks,
                        const int *kinds, const int *iparams, const double *vpar, const double *phi0,
                        const double *a0, const double *P0, int chains, uint64_t seed,
                        const uint8_t *init_gamma, int nsweeps, int lookahead, uint8_t *out_gamma,
                        double *out_beta, double *out_sigsq, double *out_variances, double *out_phi,
                        double *out_state, double *out_logpri, uint64_t *out_seed, int probe_chain,
                        uint8_t *probe_gamma, double *probe_state, int ndevices) {
  try {
    GlobalRng::rng.seed(seed);
    Matrix Xm(T, p);
    for (int j = 0; j < p; ++j)
      for (int t = 0; t < T; ++t) Xm(t, j) = X[(size_t)j * T + t];
    Vector yv(T);
    for (int t = 0; t < T; ++t) yv[t] = y[t];
    std::vector<bool> obs;
    if (observed) {
      obs.resize(T);
      for (int t = 0; t < T; ++t) obs[t] = observed[t] != 0;
    }
    NEW(StateSpaceRegressionModel, model)(yv, Xm, obs);
    RegressionModel *reg = model->observation_model();
    Vector mu(p), piv(p);
    SpdMatrix om(p);
    for (int j = 0; j < p; ++j) {
      mu[j] = prior_mean[j];
      piv[j] = pi[j];
      for (int i = 0; i < p; ++i) om(i, j) = ominv[(size_t)j * p + i];
    }
    NEW(MvnGivenScalarSigma, slab)(mu, om, reg->Sigsq_prm());
    NEW(ChisqModel, siginv_prior)(prior_df, sigma_guess);
    NEW(VariableSelectionPrior, spike)(piv);
    reg->coef().drop_all();
    for (int j = 0; j < p; ++j)
      if (init_gamma[j]) reg->coef().add(j);

    std::vector<DeviceStateVariancePrior> vpriors;
    std::vector<int> t0s;
    int first = 0;
    for (int b = 0; b < nblocks; ++b) {
      const double *vp = vpar + 8 * b;
      auto vprior = [&](int v) {
        DeviceStateVariancePrior pr;
        pr.precision_prior = new ChisqModel(vp[4 * v], vp[4 * v + 1]);
        pr.sigma_upper_limit = vp[4 * v + 2];
        vpriors.push_back(pr);
      };
      int dim = 0;
      if (kinds[b] == 1) {
        NEW(LocalLevelStateModel, level)(vp[3]);
        level->set_initial_state_mean(a0[first]);
        level->set_initial_state_variance(P0[first]);
        model->add_state(level);
        vprior(0);
        dim = 1;
      } else if (kinds[b] == 2) {
        NEW(LocalLinearTrendStateModel, llt)();
        SpdMatrix Sigma(2, 0.0);
        Sigma(0, 0) = vp[3] * vp[3];
        Sigma(1, 1) = vp[7] * vp[7];
        llt->set_Sigma(Sigma);
        Vector mean(2);
        SpdMatrix var(2, 0.0);
        for (int i = 0; i < 2; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
        llt->set_initial_state_mean(mean);
        llt->set_initial_state_variance(var);
        model->add_state(llt);
        vprior(0);
        vprior(1);
        dim = 2;
      } else if (kinds[b] == 3) {
        const int ns = iparams[3 * b];
        NEW(SeasonalStateModel, seasonal)(ns, iparams[3 * b + 1]);
        seasonal->set_time_of_first_observation(iparams[3 * b + 2]);
        t0s.push_back(iparams[3 * b + 2]);
        seasonal->set_sigsq(vp[3] * vp[3]);
        dim = ns - 1;
        Vector mean(dim);
        SpdMatrix var(dim, 0.0);
        for (int i = 0; i < dim; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
        seasonal->set_initial_state_mean(mean);
        seasonal->set_initial_state_variance(var);
        model->add_state(seasonal);
        vprior(0);
      } else if (kinds[b] == 5) {
        NEW(StaticInterceptStateModel, icpt)();
        icpt->set_initial_state_mean(a0[first]);
        icpt->set_initial_state_variance(P0[first]);
        model->add_state(icpt);
        dim = 1;   // (no parameter: no variance prior)
      } else if (kinds[b] == 7) {
        const double *pp = phi0 + 16 * b;
        NEW(ZeroMeanGaussianModel, level)(vp[3]);
        NEW(NonzeroMeanAr1Model, slope)(pp[4], pp[5], vp[7]);
        NEW(SemilocalLinearTrendStateModel, trend)(level, slope);
        trend->set_initial_level_mean(a0[first]);
        trend->set_initial_slope_mean(a0[first + 1]);
        trend->set_initial_level_sd(std::sqrt(P0[first]));
        trend->set_initial_slope_sd(std::sqrt(P0[first + 1]));
        model->add_state(trend);
        vprior(0);
        vprior(1);
        vpriors.back().slope_mean_prior = new GaussianModel(pp[0], pp[1]);
        vpriors.back().slope_ar1_prior = new GaussianModel(pp[2], pp[3]);
        vpriors.back().force_stationary = iparams[3 * b] != 0;
        vpriors.back().force_ar1_positive = iparams[3 * b + 1] != 0;
        dim = 3;
      } else if (kinds[b] == 6) {
        const int nf = iparams[3 * b];
        Vector freqs(nf);
        for (int i = 0; i < nf; ++i) freqs[i] = phi0[16 * b + 1 + i];
        NEW(TrigStateModel, trig)(phi0[16 * b], freqs);
        trig->error_distribution()->set_sigsq(vp[3] * vp[3]);
        dim = 2 * nf;
        Vector mean(dim);
        SpdMatrix var(dim, 0.0);
        for (int i = 0; i < dim; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
        trig->set_initial_state_mean(mean);
        trig->set_initial_state_variance(var);
        model->add_state(trig);
        vprior(0);
      } else {
        dim = iparams[3 * b];
        NEW(ArStateModel, arm)(dim);
        Vector ph(dim);
        for (int i = 0; i < dim; ++i) ph[i] = phi0[16 * b + i];
        arm->set_phi(ph);
        arm->set_sigma(vp[3]);
        Vector mean(dim);
        SpdMatrix var(dim, 0.0);
        for (int i = 0; i < dim; ++i) { mean[i] = a0[first + i]; var(i, i) = P0[first + i]; }
        arm->set_initial_state_mean(mean);
        arm->set_initial_state_variance(var);
        model->add_state(arm);
        vprior(0);
      }
      first += dim;
    }
    const int m = first;
    // (ndevices > 0: `chains` per entry of a device list that names device 0 ndevices times)
    Ptr<DeviceStateSpacePosteriorSampler> sampler(
        ndevices > 0 ? new DeviceStateSpacePosteriorSampler(model.get(), slab, siginv_prior, spike, sigma_upper_limit,
                                                            vpriors, chains, std::vector<int>(ndevices, 0),
                                                            GlobalRng::rng, t0s, lookahead)
                     : new DeviceStateSpacePosteriorSampler(model.get(), slab, siginv_prior, spike, sigma_upper_limit,
                                                            vpriors, chains, 0, GlobalRng::rng, t0s, lookahead));
    if (out_seed) *out_seed = sampler->device_seed();
    model->set_method(sampler);
    for (int s = 0; s < nsweeps; ++s) {
      model->sample_posterior();   // PriorPolicy::sample_posterior -> sampler->draw()
      const Selector &inc(reg->coef().inc());
      const Vector beta = reg->Beta();
      for (int j = 0; j < p; ++j) {
        out_gamma[(size_t)s * p + j] = inc[j] ? 1 : 0;
        out_beta[(size_t)s * p + j] = beta[j];
      }
      out_sigsq[s] = reg->sigsq();
      for (int b = 0; b < nblocks; ++b) {
        double *v = out_variances + ((size_t)s * nblocks + b) * 2;
        double *ph = out_phi + ((size_t)s * nblocks + b) * 16;
        v[0] = v[1] = 0.0;
        for (int i = 0; i < 16; ++i) ph[i] = 0.0;
        StateModel *sm = model->state_model(b);
        if (kinds[b] == 1) {
          v[0] = dynamic_cast<LocalLevelStateModel *>(sm)->sigsq();
        } else if (kinds[b] == 2) {
          const SpdMatrix S = dynamic_cast<LocalLinearTrendStateModel *>(sm)->Sigma();
          v[0] = S(0, 0);
          v[1] = S(1, 1);
        } else if (kinds[b] == 3) {
          v[0] = dynamic_cast<SeasonalStateModel *>(sm)->sigsq();
        } else if (kinds[b] == 5) {
          // (no parameter)
        } else if (kinds[b] == 6) {
          v[0] = dynamic_cast<TrigStateModel *>(sm)->error_distribution()->sigsq();
        } else if (kinds[b] == 7) {
          SemilocalLinearTrendStateModel *trend = dynamic_cast<SemilocalLinearTrendStateModel *>(sm);
          v[0] = trend->level_sd() * trend->level_sd();
          v[1] = trend->slope_sd() * trend->slope_sd();
          ph[0] = trend->slope_ar_coefficient();
          ph[1] = trend->slope_mean();
        } else {
          ArStateModel *arm = dynamic_cast<ArStateModel *>(sm);
          v[0] = arm->sigsq();
          for (int i = 0; i < iparams[3 * b]; ++i) ph[i] = arm->phi()[i];
        }
      }
      const Matrix &state(model->state());
      if (state.nrow() != m || state.ncol() != T) throw std::runtime_error("state has the wrong shape");
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < m; ++i) out_state[((size_t)s * T + t) * m + i] = state(i, t);
      if (out_logpri) out_logpri[s] = sampler->logpri();
    }
    if (probe_gamma) {
      Selector inc(p, false);
      Vector beta, variances;
      Matrix state;
      double s2 = 0;
      sampler->chain_state(probe_chain, inc, beta, s2, variances, state);
      for (int j = 0; j < p; ++j) probe_gamma[j] = inc[j] ? 1 : 0;
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < m; ++i) probe_state[(size_t)t * m + i] = state(i, t);
    }
    return 0;
  } catch (std::exception &e) {
    g_binding_error = e.what();
    return -1;
  }
}

// ... and for the Poisson sampler: BOOM's PoissonRegressionModel (one PoissonRegressionData
// per observation), MvnModel slab, VariableSelectionPrior, sample_posterior() with the
// device sampler attached -- which reads BOOM's own NegLogGamma mixture table.
int ref_binding_poisson_run(int n, int p, const double *X, const double *y, const double *exposure,
                            const double *slab_mean, const double *slab_precision, const double *pi,
                            int max_flips, int chains, uint64_t seed, const uint8_t *init_gamma,
                            int nsweeps, uint8_t *out_gamma, double *out_beta, uint64_t *out_seed) {
  try {
    GlobalRng::rng.seed(seed);
    Ptr<PoissonRegressionModel> model(new PoissonRegressionModel(p));
    for (int i = 0; i < n; ++i) {
      Vector x(p);
      for (int j = 0; j < p; ++j) x[j] = X[(size_t)j * n + i];
      NEW(PoissonRegressionData, dp)((int64_t)llround(y[i]), x, exposure[i]);
      model->add_data(dp);
    }
    Vector mu(p), piv(p);
    SpdMatrix prec(p);
    for (int j = 0; j < p; ++j) {
      mu[j] = slab_mean[j];
      piv[j] = pi[j];
      for (int i = 0; i < p; ++i) prec(i, j) = slab_precision[(size_t)j * p + i];
    }
    NEW(MvnModel, slab)(mu, prec, true);
    NEW(VariableSelectionPrior, spike)(piv);
    model->coef().drop_all();
    for (int j = 0; j < p; ++j)
      if (init_gamma[j]) model->coef().add(j);
    NEW(DevicePoissonRegressionSpikeSlabSampler, sampler)(model.get(), slab, spike, chains);
    if (max_flips > 0) sampler->limit_model_selection(max_flips);
    if (out_seed) *out_seed = sampler->device_seed();
    model->set_method(sampler);
    for (int s = 0; s < nsweeps; ++s) {
      model->sample_posterior();
      const Selector &inc(model->coef().inc());
      const Vector beta = model->Beta();
      for (int j = 0; j < p; ++j) {
        out_gamma[(size_t)s * p + j] = inc[j] ? 1 : 0;
        out_beta[(size_t)s * p + j] = beta[j];
      }
    }
    return 0;
  } catch (std::exception &e) {
    g_binding_error = e.what();
    return -1;
  }
}

}  // extern "C"
