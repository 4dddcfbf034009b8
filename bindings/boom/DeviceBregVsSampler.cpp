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
        group_(nullptr),
        chains_(chains),
        max_flips_(-1),
        swap_threshold_(0.8) {
    device_seed_ = seed_rng(seeding_rng);
    ba_config cfg{device, chains, 0, static_cast<uint64_t>(device_seed_), 0, 0};
    ba_engine *engine = nullptr;
    check(ba_engine_create(&cfg, &engine));
    engines_.push_back(engine);
    try {   // (report_error throws and a throwing constructor runs no destructor: nothing may leak)
      configure(slab, residual_precision_prior, spike, lookahead);
    } catch (...) {
      ba_engine_destroy(engine);
      engines_.clear();
      throw;
    }
  }

  DeviceBregVsSampler::DeviceBregVsSampler(
      RegressionModel *model, const Ptr<MvnGivenScalarSigmaBase> &slab,
      const Ptr<GammaModelBase> &residual_precision_prior,
      const Ptr<VariableSelectionPrior> &spike, int chains_per_device,
      const std::vector<int> &devices, int lookahead, RNG &seeding_rng)
      : PosteriorSampler(seeding_rng),
        model_(model),
        group_(nullptr),
        chains_(chains_per_device * static_cast<int>(devices.size())),
        max_flips_(-1),
        swap_threshold_(0.8) {
    device_seed_ = seed_rng(seeding_rng);
    std::vector<int32_t> dev(devices.begin(), devices.end());
    if (ba_group_create(dev.data(), static_cast<int32_t>(dev.size()), chains_per_device,
                        static_cast<uint64_t>(device_seed_), &group_) != BA_OK) {
      report_error(ba_group_last_error());
    }
    for (int32_t i = 0; i < ba_group_size(group_); ++i) {
      engines_.push_back(ba_group_engine(group_, i));
    }
    try {
      configure(slab, residual_precision_prior, spike, lookahead);
    } catch (...) {
      ba_group_destroy(group_);
      group_ = nullptr;
      engines_.clear();
      throw;
    }
  }

  void DeviceBregVsSampler::configure(
      const Ptr<MvnGivenScalarSigmaBase> &slab,
      const Ptr<GammaModelBase> &residual_precision_prior,
      const Ptr<VariableSelectionPrior> &spike, int lookahead) {
    if (slab->dim() != static_cast<int>(model_->xdim())) {
      report_error("Slab dimension did not match model dimension.");
    }
    if (spike->potential_nvars() != model_->xdim()) {
      report_error("Spike dimension did not match model dimension.");
    }
    Ptr<RegSuf> suf = model_->suf();
    const SpdMatrix xtx = suf->xtx();  // column-major, full storage
    const Vector xty = suf->xty();
    const Vector xbar = suf->xbar();
    const Vector mu = slab->mu();
    const SpdMatrix ominv = slab->unscaled_precision();
    const Vector pi = spike->prior_inclusion_probabilities();
    // GammaModel(alpha, beta) == ChisqModel(df = 2 alpha, sigma = sqrt(beta / alpha))
    const double a = residual_precision_prior->alpha();
    const double b = residual_precision_prior->beta();
    prior_df_ = 2 * a;
    prior_sigma_guess_ = std::sqrt(b / a);
    for (ba_engine *engine : engines_) {
      check(ba_upload_regression_suf(engine, xtx.nrow(), xtx.data(), xty.data(),
                                     suf->yty(), suf->n(), suf->ybar(), xbar.data()));
      check(ba_set_slab(engine, mu.data(), ominv.data()));
      check(ba_set_spike(engine, pi.data(), spike->max_model_size()));
      check(ba_set_sigma_prior(engine, prior_df_, prior_sigma_guess_, infinity()));
    }
    push_state();
    if (lookahead > 1) set_lookahead(lookahead);
  }

  DeviceBregVsSampler::~DeviceBregVsSampler() {
    if (group_) {
      ba_group_destroy(group_);
    } else {
      for (ba_engine *engine : engines_) ba_engine_destroy(engine);
    }
  }

  void DeviceBregVsSampler::check(int rc) const {
    if (rc != BA_OK) report_error(ba_last_error());
  }

  void DeviceBregVsSampler::draw() {
    // every engine is asked first (the launches are asynchronous, so the devices
    // run side by side); chain 0 is then read from engine 0
    for (ba_engine *engine : engines_) check(ba_draw_next(engine));
    pull_chain0();
  }

  double DeviceBregVsSampler::logpri() const {
    double ans = negative_infinity();
    check(ba_logpri(engines_[0], 0, &ans));
    return ans;
  }

  void DeviceBregVsSampler::set_device_seed(unsigned long seed) {
    device_seed_ = seed;
    for (ba_engine *engine : engines_) check(ba_seed(engine, seed));
  }

  void DeviceBregVsSampler::options() {
    for (ba_engine *engine : engines_) {
      check(ba_set_options(engine, max_flips_, swap_threshold_, 1, 1));
    }
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
    for (ba_engine *engine : engines_) {
      check(ba_set_sigma_prior(engine, prior_df_, prior_sigma_guess_, sigma_upper_limit));
    }
  }
  void DeviceBregVsSampler::set_lookahead(int n) {
    for (ba_engine *engine : engines_) check(ba_set_lookahead(engine, n));
  }

  void DeviceBregVsSampler::push_state() {
    const Selector &inc(model_->coef().inc());
    const uint p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    for (uint j = 0; j < p; ++j) gamma[j] = inc[j] ? 1 : 0;
    const Vector beta = model_->Beta();
    for (ba_engine *engine : engines_) {
      check(ba_set_state(engine, -1, gamma.data(), beta.data(), model_->sigsq()));
    }
  }

  void DeviceBregVsSampler::chain_state(int chain, Selector &inc, Vector &beta,
                                        double &sigsq) const {
    const uint p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    beta.resize(p);
    int32_t owner = 0;
    int64_t local = chain;
    if (group_ && ba_group_locate(group_, chain, &owner, &local) != BA_OK) {
      report_error(ba_group_last_error());
    }
    check(ba_get_state(engines_[owner], local, gamma.data(), beta.data(), &sigsq));
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
