// See DeviceBregVsSampler.hpp.  OUR code against the reference's headers.
#include "DeviceBregVsSampler.hpp"

#include <cmath>
#include <vector>

#include "cpputil/math_utils.hpp"
#include "cpputil/report_error.hpp"
#include "distributions/rng.hpp"
#include "Models/ChisqModel.hpp"

namespace BOOM {

  // ---- construction ---------------------------------------------------------------
  void DeviceBregVsSampler::create_engine(int device, RNG &seeding_rng) {
    device_seed_ = seed_rng(seeding_rng);
    ba_config cfg{device, chains_, 0, static_cast<uint64_t>(device_seed_), 0, 0};
    ba_engine *engine = nullptr;
    check(ba_engine_create(&cfg, &engine));
    engines_.push_back(engine);
  }

  void DeviceBregVsSampler::destroy_engines() {
    if (group_) {
      ba_group_destroy(group_);
      group_ = nullptr;
    } else {
      for (ba_engine *engine : engines_) ba_engine_destroy(engine);
    }
    engines_.clear();
  }

  void DeviceBregVsSampler::upload_suf() {
    Ptr<RegSuf> suf = model_->suf();
    const SpdMatrix xtx = suf->xtx();  // column-major, full storage
    const Vector xty = suf->xty();
    const Vector xbar = suf->xbar();
    for (ba_engine *engine : engines_) {
      check(ba_upload_regression_suf(engine, xtx.nrow(), xtx.data(), xty.data(),
                                     suf->yty(), suf->n(), suf->ybar(), xbar.data()));
    }
  }

  // (#1, #2) engine 0 has assembled the priors: the sampler's objects hold the same numbers
  void DeviceBregVsSampler::priors_from_engine() {
    const uint p = model_->xdim();
    Vector mu(p), pi(p);
    SpdMatrix ominv(p);
    double df = 0, ss = 0;
    check(ba_get_priors(engines_[0], mu.data(), ominv.data(), pi.data(), &df, &ss));
    slab_ = new MvnGivenScalarSigma(mu, ominv, model_->Sigsq_prm());
    residual_precision_prior_ = new ChisqModel(df, std::sqrt(ss / df));
    spike_ = new VariableSelectionPrior(pi);
  }

#define BOOM_AMD_SINGLE_DEVICE_INIT                                              \
      : PosteriorSampler(seeding_rng), model_(model), group_(nullptr), chains_(chains), \
        max_flips_(-1), swap_threshold_(0.8), sigma_upper_limit_(infinity()), priors_stale_(false)

  // #1
  DeviceBregVsSampler::DeviceBregVsSampler(RegressionModel *model, double prior_nobs,
                                           double expected_rsq, double expected_model_size,
                                           bool first_term_is_intercept, int chains, int device,
                                           int lookahead, RNG &seeding_rng)
      BOOM_AMD_SINGLE_DEVICE_INIT {
    create_engine(device, seeding_rng);
    try {
      upload_suf();
      check(ba_set_priors_ctor1(engines_[0], prior_nobs, expected_rsq, expected_model_size,
                                first_term_is_intercept ? 1 : 0));
      priors_from_engine();
      configure(lookahead);
    } catch (...) {   // (report_error throws and a throwing constructor runs no destructor)
      destroy_engines();
      throw;
    }
  }

  // #2
  DeviceBregVsSampler::DeviceBregVsSampler(RegressionModel *model, double prior_sigma_nobs,
                                           double prior_sigma_guess, double prior_beta_nobs,
                                           double diagonal_shrinkage,
                                           double prior_inclusion_probability, bool force_intercept,
                                           int chains, int device, int lookahead, RNG &seeding_rng)
      BOOM_AMD_SINGLE_DEVICE_INIT {
    create_engine(device, seeding_rng);
    try {
      upload_suf();
      check(ba_set_priors_ctor2(engines_[0], prior_sigma_nobs, prior_sigma_guess, prior_beta_nobs,
                                diagonal_shrinkage, prior_inclusion_probability,
                                force_intercept ? 1 : 0));
      priors_from_engine();
      configure(lookahead);
    } catch (...) {
      destroy_engines();
      throw;
    }
  }

  // #3
  DeviceBregVsSampler::DeviceBregVsSampler(RegressionModel *model, const Vector &prior_mean,
                                           const SpdMatrix &unscaled_prior_precision,
                                           double sigma_guess, double df,
                                           const Vector &prior_inclusion_probs, int chains,
                                           int device, int lookahead, RNG &seeding_rng)
      BOOM_AMD_SINGLE_DEVICE_INIT {
    slab_ = new MvnGivenScalarSigma(prior_mean, unscaled_prior_precision, model_->Sigsq_prm());
    residual_precision_prior_ = new ChisqModel(df, sigma_guess);
    spike_ = new VariableSelectionPrior(prior_inclusion_probs);
    create_engine(device, seeding_rng);
    try {
      upload_suf();
      configure(lookahead);
    } catch (...) {
      destroy_engines();
      throw;
    }
  }

  // #4
  DeviceBregVsSampler::DeviceBregVsSampler(RegressionModel *model,
                                           const ZellnerPriorParameters &prior, int chains,
                                           int device, int lookahead, RNG &seeding_rng)
      BOOM_AMD_SINGLE_DEVICE_INIT {
    slab_ = new MvnGivenScalarSigma(prior.prior_beta_guess, prior.prior_beta_information,
                                    model_->Sigsq_prm());
    residual_precision_prior_ =
        new ChisqModel(prior.prior_sigma_guess_weight, prior.prior_sigma_guess);
    spike_ = new VariableSelectionPrior(prior.prior_inclusion_probabilities);
    create_engine(device, seeding_rng);
    try {
      upload_suf();
      configure(lookahead);
    } catch (...) {
      destroy_engines();
      throw;
    }
  }

  // #5
  DeviceBregVsSampler::DeviceBregVsSampler(
      RegressionModel *model, const Ptr<MvnGivenScalarSigmaBase> &slab,
      const Ptr<GammaModelBase> &residual_precision_prior,
      const Ptr<VariableSelectionPrior> &spike, int chains, int device,
      int lookahead, RNG &seeding_rng)
      BOOM_AMD_SINGLE_DEVICE_INIT {
    slab_ = slab;
    residual_precision_prior_ = residual_precision_prior;
    spike_ = spike;
    create_engine(device, seeding_rng);
    try {
      upload_suf();
      configure(lookahead);
    } catch (...) {
      destroy_engines();
      throw;
    }
  }
#undef BOOM_AMD_SINGLE_DEVICE_INIT

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
        swap_threshold_(0.8),
        sigma_upper_limit_(infinity()),
        slab_(slab),
        residual_precision_prior_(residual_precision_prior),
        spike_(spike),
        priors_stale_(false) {
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
      upload_suf();
      configure(lookahead);
    } catch (...) {
      destroy_engines();
      throw;
    }
  }

  // slab, spike, residual prior -> every engine (BregVsSampler reads them at every draw:
  // set_reg_post_params, BregVsSampler.cpp:395-484; log_model_prob :216-239)
  void DeviceBregVsSampler::upload_priors() {
    if (slab_->dim() != static_cast<int>(model_->xdim())) {
      report_error("Slab dimension did not match model dimension.");
    }
    if (spike_->potential_nvars() != model_->xdim()) {
      report_error("Spike dimension did not match model dimension.");
    }
    const Vector mu = slab_->mu();
    const SpdMatrix ominv = slab_->unscaled_precision();
    const Vector pi = spike_->prior_inclusion_probabilities();
    // GammaModel(alpha, beta) == ChisqModel(df = 2 alpha, sigma = sqrt(beta / alpha))
    const double a = residual_precision_prior_->alpha();
    const double b = residual_precision_prior_->beta();
    prior_df_ = 2 * a;
    prior_sigma_guess_ = std::sqrt(b / a);
    for (ba_engine *engine : engines_) {
      check(ba_set_slab(engine, mu.data(), ominv.data()));
      check(ba_set_spike(engine, pi.data(), spike_->max_model_size()));
      check(ba_set_sigma_prior(engine, prior_df_, prior_sigma_guess_, sigma_upper_limit_));
    }
    priors_stale_ = false;
  }

  void DeviceBregVsSampler::configure(int lookahead) {
    upload_priors();
    observe();
    push_state();
    if (lookahead > 1) set_lookahead(lookahead);
  }

  // Data::add_observer (DataTypes.hpp:76) on every parameter of the three prior models:
  // a set() on any of them marks the device copies stale
  void DeviceBregVsSampler::observe() {
    auto watch = [this](Model *m) {
      for (Ptr<Params> &prm : m->parameter_vector()) {
        prm->add_observer(this, [this]() { this->priors_stale_ = true; });
      }
    };
    watch(slab_.get());
    watch(residual_precision_prior_.get());
    watch(spike_.get());
  }
  void DeviceBregVsSampler::unobserve() {
    auto unwatch = [this](Model *m) {
      for (Ptr<Params> &prm : m->parameter_vector()) prm->remove_observer(this);
    };
    if (!!slab_) unwatch(slab_.get());
    if (!!residual_precision_prior_) unwatch(residual_precision_prior_.get());
    if (!!spike_) unwatch(spike_.get());
  }

  DeviceBregVsSampler::~DeviceBregVsSampler() {
    unobserve();
    destroy_engines();
  }

  void DeviceBregVsSampler::check(int rc) const {
    if (rc != BA_OK) report_error(ba_last_error());
  }

  void DeviceBregVsSampler::draw() {
    // (a prior that changed since the last draw: the engine puts the chains back at the
    // draw the caller has seen before it takes the new values -- ba_set_slab and the others
    // are mutators behind the look-ahead)
    if (priors_stale_) upload_priors();
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
    sigma_upper_limit_ = sigma_upper_limit;
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
