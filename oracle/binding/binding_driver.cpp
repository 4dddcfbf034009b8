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
#include "Models/MvnModel.hpp"
#include "LinAlg/Matrix.hpp"
#include "LinAlg/SpdMatrix.hpp"
#include "Models/ChisqModel.hpp"
#include "distributions/rng.hpp"

using namespace BOOM;

static std::string g_binding_error;

extern "C" {

const char *ref_binding_last_error() { return g_binding_error.c_str(); }

// X n x p column-major.  Returns 0 or -1 (message in ref_binding_last_error).
// out_seed receives the seed the binding gave the engine (seed_rng(GlobalRng)
// after GlobalRng::rng.seed(seed)), i.e. the key of the oracle run to compare with.
int ref_binding_run(int n, int p, const double *X, const double *y,
                    const double *prior_mean, const double *ominv, double prior_df,
                    double sigma_guess, const double *pi, int64_t max_model_size,
                    double sigma_upper_limit, int max_flips, double swap_threshold,
                    int chains, int lookahead, uint64_t seed, const uint8_t *init_gamma,
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
    NEW(DeviceBregVsSampler, sampler)(model.get(), slab, siginv_prior, spike, chains, 0,
                                      lookahead);
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

// The same for the logit sampler: BOOM's BinomialLogitModel (data added one
// BinomialRegressionData at a time, as the reference's callers do), MvnModel slab,
// VariableSelectionPrior, and sample_posterior() with the device sampler attached.
int ref_binding_logit_run(int n, int p, const double *X, const double *y, const double *ntrials,
                          const double *slab_mean, const double *slab_precision, const double *pi,
                          int clt_threshold, int max_flips, int chains, uint64_t seed,
                          const uint8_t *init_gamma, int nsweeps, uint8_t *out_gamma,
                          double *out_beta, uint64_t *out_seed, int probe_chain,
                          uint8_t *probe_gamma, double *probe_beta) {
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
    NEW(DeviceBinomialLogitSpikeSlabSampler, sampler)(model.get(), slab, spike, clt_threshold, chains);
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

}  // extern "C"
