// See DeviceBregVsSampler.hpp.  OUR code against the reference's headers.
#include "DeviceBregVsSampler.hpp"

#include <cmath>
#include <vector>

#include "cpputil/math_utils.hpp"
#include "cpputil/report_error.hpp"
#include "distributions/rng.hpp"

namespace BOOM {

  DeviceBregVsSampler::DeviceBregVsSampler(
      RegressionModel *model, const Ptr<MvnGivenScalarSigmaBase> &slab,
      const Ptr<GammaModelBase> &residual_precision_prior,
      const Ptr<VariableSelectionPrior> &spike, int chains, int device,
      int lookahead, RNG &seeding_rng)
      : PosteriorSampler(seeding_rng),
        model_(model),
        engine_(nullptr),
        chains_(chains),
        max_flips_(-1),
        swap_threshold_(0.8) {
    if (slab->dim() != static_cast<int>(model->xdim())) {
      report_error("Slab dimension did not match model dimension.");
    }
    if (spike->potential_nvars() != model->xdim()) {
      report_error("Spike dimension did not match model dimension.");
    }
    device_seed_ = seed_rng(seeding_rng);
    ba_config cfg{device, chains, 0, static_cast<uint64_t>(device_seed_), 0, 0};
    check(ba_engine_create(&cfg, &engine_));
    Ptr<RegSuf> suf = model->suf();
    const SpdMatrix xtx = suf->xtx();  // column-major, full storage
    const Vector xty = suf->xty();
    const Vector xbar = suf->xbar();
    check(ba_upload_regression_suf(engine_, xtx.nrow(), xtx.data(), xty.data(),
                                   suf->yty(), suf->n(), suf->ybar(), xbar.data()));
    const Vector mu = slab->mu();
    const SpdMatrix ominv = slab->unscaled_precision();
    check(ba_set_slab(engine_, mu.data(), ominv.data()));
    const Vector pi = spike->prior_inclusion_probabilities();
    check(ba_set_spike(engine_, pi.data(), spike->max_model_size()));
    // GammaModel(alpha, beta) == ChisqModel(df = 2 alpha, sigma = sqrt(beta / alpha))
    const double a = residual_precision_prior->alpha();
    const double b = residual_precision_prior->beta();
    prior_df_ = 2 * a;
    prior_sigma_guess_ = std::sqrt(b / a);
    check(ba_set_sigma_prior(engine_, prior_df_, prior_sigma_guess_, infinity()));
    push_state();
    if (lookahead > 1) check(ba_set_lookahead(engine_, lookahead));
  }

  DeviceBregVsSampler::~DeviceBregVsSampler() { ba_engine_destroy(engine_); }

  void DeviceBregVsSampler::check(int rc) const {
    if (rc != BA_OK) report_error(ba_last_error());
  }

  void DeviceBregVsSampler::draw() {
    check(ba_draw_next(engine_));
    pull_chain0();
  }

  double DeviceBregVsSampler::logpri() const {
    double ans = negative_infinity();
    check(ba_logpri(engine_, 0, &ans));
    return ans;
  }

  void DeviceBregVsSampler::set_device_seed(unsigned long seed) {
    device_seed_ = seed;
    check(ba_seed(engine_, seed));
  }

  void DeviceBregVsSampler::options() {
    check(ba_set_options(engine_, max_flips_, swap_threshold_, 1, 1));
  }
  void DeviceBregVsSampler::limit_model_selection(uint max_flips) {
    max_flips_ = static_cast<int>(max_flips);
    options();
  }
  void DeviceBregVsSampler::suppress_model_selection() {
    max_flips_ = 0;
    options();
  }
  void DeviceBregVsSampler::allow_model_selection() {
    max_flips_ = -1;
    options();
  }
  void DeviceBregVsSampler::set_correlation_swap_threshold(double threshold) {
    swap_threshold_ = threshold;
    options();
  }
  void DeviceBregVsSampler::set_sigma_upper_limit(double sigma_upper_limit) {
    check(ba_set_sigma_prior(engine_, prior_df_, prior_sigma_guess_, sigma_upper_limit));
  }
  void DeviceBregVsSampler::set_lookahead(int n) {
    check(ba_set_lookahead(engine_, n));
  }

  void DeviceBregVsSampler::push_state() {
    const Selector &inc(model_->coef().inc());
    const uint p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    for (uint j = 0; j < p; ++j) gamma[j] = inc[j] ? 1 : 0;
    const Vector beta = model_->Beta();
    check(ba_set_state(engine_, -1, gamma.data(), beta.data(), model_->sigsq()));
  }

  void DeviceBregVsSampler::chain_state(int chain, Selector &inc, Vector &beta,
                                        double &sigsq) const {
    const uint p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    beta.resize(p);
    check(ba_get_state(engine_, chain, gamma.data(), beta.data(), &sigsq));
    inc = Selector(p, false);
    for (uint j = 0; j < p; ++j) {
      if (gamma[j]) inc.add(j);
    }
  }

  void DeviceBregVsSampler::pull_chain0() {
    Selector inc(model_->xdim(), false);
    Vector beta;
    double sigsq = 1.0;
    chain_state(0, inc, beta, sigsq);
    model_->coef().set_inc(inc);
    model_->set_included_coefficients(inc.select(beta));
    model_->set_sigsq(sigsq);
  }

}  // namespace BOOM
