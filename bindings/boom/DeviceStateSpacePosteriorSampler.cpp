// See DeviceStateSpacePosteriorSampler.hpp.  OUR code against the reference's headers.
#include "DeviceStateSpacePosteriorSampler.hpp"

#include <cmath>
#include <cstring>

#include "Models/StateSpace/StateModels/LocalLevelStateModel.hpp"
#include "Models/StateSpace/StateModels/LocalLinearTrend.hpp"
#include "Models/StateSpace/StateModels/SeasonalStateModel.hpp"
#include "Models/StateSpace/StateModels/StaticInterceptStateModel.hpp"
#include "Models/StateSpace/StateModels/SemilocalLinearTrend.hpp"
#include "Models/StateSpace/StateModels/TrigStateModel.hpp"
#include "cpputil/math_utils.hpp"
#include "cpputil/report_error.hpp"
#include "distributions/rng.hpp"

namespace BOOM {

  namespace {
    // GammaModel(alpha, beta) == ChisqModel(df = 2 alpha, sigma = sqrt(beta / alpha))
    // (ChisqModel.cpp:56-57)
    double prior_df(const Ptr<GammaModelBase> &prior) { return 2 * prior->alpha(); }
    double prior_sigma_guess(const Ptr<GammaModelBase> &prior) {
      return std::sqrt(prior->beta() / prior->alpha());
    }
    // the engine takes independent initial state components
    void diagonal_or_die(const SpdMatrix &V, const char *what) {
      for (int i = 0; i < V.nrow(); ++i)
        for (int j = 0; j < i; ++j)
          if (V(i, j) != 0.0)
            report_error(std::string("the device sampler needs a diagonal initial state variance for ") + what);
    }
  }  // namespace

  DeviceStateSpacePosteriorSampler::DeviceStateSpacePosteriorSampler(
      StateSpaceRegressionModel *model, const Ptr<MvnGivenScalarSigmaBase> &slab,
      const Ptr<GammaModelBase> &residual_precision_prior,
      const Ptr<VariableSelectionPrior> &spike, double sigma_upper_limit,
      const std::vector<DeviceStateVariancePrior> &state_variance_priors, int chains,
      int device, RNG &seeding_rng, const std::vector<int> &seasonal_time_of_first_observation,
      int lookahead)
      : PosteriorSampler(seeding_rng),
        model_(model),
        engine_(nullptr),
        chains_(chains),
        state_dim_(0),
        structural_(false),
        variance_priors_(state_variance_priors) {
    classify(slab, spike);
    device_seed_ = seed_rng(seeding_rng);
    ba_config cfg{device, chains, 0, static_cast<uint64_t>(device_seed_), 0, 0};
    check(ba_engine_create(&cfg, &engine_));
    engines_.assign(1, engine_);
    // (from here on a failure must not leak the engine: report_error throws)
    try {
      configure(slab, residual_precision_prior, spike, sigma_upper_limit, seasonal_time_of_first_observation,
                lookahead);
    } catch (...) {
      ba_engine_destroy(engine_);
      engine_ = nullptr;
      throw;
    }
  }

  DeviceStateSpacePosteriorSampler::DeviceStateSpacePosteriorSampler(
      StateSpaceRegressionModel *model, const Ptr<MvnGivenScalarSigmaBase> &slab,
      const Ptr<GammaModelBase> &residual_precision_prior,
      const Ptr<VariableSelectionPrior> &spike, double sigma_upper_limit,
      const std::vector<DeviceStateVariancePrior> &state_variance_priors, int chains_per_device,
      const std::vector<int> &devices, RNG &seeding_rng,
      const std::vector<int> &seasonal_time_of_first_observation, int lookahead)
      : PosteriorSampler(seeding_rng),
        model_(model),
        engine_(nullptr),
        chains_(chains_per_device * static_cast<int>(devices.size())),
        state_dim_(0),
        structural_(false),
        variance_priors_(state_variance_priors) {
    if (devices.empty()) report_error("The device list is empty.");
    classify(slab, spike);
    device_seed_ = seed_rng(seeding_rng);
    std::vector<int32_t> dv(devices.begin(), devices.end());
    if (ba_group_create(dv.data(), static_cast<int32_t>(dv.size()), chains_per_device,
                        static_cast<uint64_t>(device_seed_), &group_) != BA_OK)
      report_error(ba_group_last_error());
    for (int i = 0; i < ba_group_size(group_); ++i) engines_.push_back(ba_group_engine(group_, i));
    engine_ = engines_[0];
    try {
      configure(slab, residual_precision_prior, spike, sigma_upper_limit, seasonal_time_of_first_observation,
                lookahead);
    } catch (...) {
      ba_group_destroy(group_);
      group_ = nullptr;
      engine_ = nullptr;
      throw;
    }
  }

  // which state models, in the order add_state received them
  void DeviceStateSpacePosteriorSampler::classify(const Ptr<MvnGivenScalarSigmaBase> &slab,
                                                  const Ptr<VariableSelectionPrior> &spike) {
    StateSpaceRegressionModel *model = model_;
    const int p = model->xdim();
    const int T = model->time_dimension();
    if (slab->dim() != p) report_error("Slab dimension did not match model dimension.");
    if (static_cast<int>(spike->potential_nvars()) != p)
      report_error("Spike dimension did not match model dimension.");
    if (T <= 0) report_error("Add the data to the model before creating the device sampler.");

    // ---- which state models, in the order add_state received them
    const int nstate = model->number_of_state_models();
    if (nstate < 1) report_error("No state has been defined.");
    int nvar = 0;
    for (int s = 0; s < nstate; ++s) {
      const StateModel *sm = model->state_model(s);
      Block b{0, nvar, 1, 0, 0};
      if (dynamic_cast<const LocalLevelStateModel *>(sm)) {
        b.kind = 1; b.dim = 1;
      } else if (dynamic_cast<const LocalLinearTrendStateModel *>(sm)) {
        b.kind = 2; b.dim = 2; b.nvar = 2;
      } else if (const SeasonalStateModel *seas = dynamic_cast<const SeasonalStateModel *>(sm)) {
        b.kind = 3; b.dim = seas->nseasons() - 1;
      } else if (const ArStateModel *ar = dynamic_cast<const ArStateModel *>(sm)) {
        b.kind = 4; b.lags = ar->number_of_lags(); b.dim = b.lags;
      } else if (dynamic_cast<const StaticInterceptStateModel *>(sm)) {
        b.kind = 5; b.dim = 1; b.nvar = 0;   // (nothing to learn: no entry in state_variance_priors)
      } else if (dynamic_cast<const TrigStateModel *>(sm)) {
        b.kind = 6; b.dim = static_cast<int>(sm->state_dimension());   // (one variance for all 2 x frequencies components)
      } else if (dynamic_cast<const SemilocalLinearTrendStateModel *>(sm)) {
        b.kind = 7; b.dim = 3; b.nvar = 2;   // (level, slope; the slope's entry carries the NonzeroMeanAr1Sampler's other priors)
      } else {
        report_error("The device sampler takes LocalLevelStateModel, LocalLinearTrendStateModel, "
                     "SeasonalStateModel, ArStateModel, StaticInterceptStateModel, TrigStateModel and "
                     "SemilocalLinearTrendStateModel state models.");
      }
      nvar += b.nvar;
      state_dim_ += b.dim;
      blocks_.push_back(b);
    }
    structural_ = !(nstate == 1 && blocks_[0].kind == 1);
    if (variance_priors_.size() != static_cast<size_t>(nvar))
      report_error("state_variance_priors needs one entry per state variance parameter.");

  }

  void DeviceStateSpacePosteriorSampler::configure(
      const Ptr<MvnGivenScalarSigmaBase> &slab, const Ptr<GammaModelBase> &residual_precision_prior,
      const Ptr<VariableSelectionPrior> &spike, double sigma_upper_limit,
      const std::vector<int> &seasonal_time_of_first_observation, int lookahead) {
    StateSpaceRegressionModel *model = model_;
    const int p = model->xdim();
    const int T = model->time_dimension();
    const int nstate = model->number_of_state_models();
    // ---- data: one RegressionData per time point (StateSpaceRegressionModel.cpp:100-125)
    Vector y(T, 0.0);
    Matrix X(T, p);
    std::vector<uint8_t> observed(T, 1);
    for (int t = 0; t < T; ++t) {
      const Ptr<StateSpace::MultiplexedRegressionData> &dp(model->dat()[t]);
      if (dp->total_sample_size() != 1)
        report_error("The device sampler takes one observation per time point.");
      const RegressionData &rd(dp->regression_data(0));
      X.row(t) = rd.x();
      if (model->is_missing_observation(t)) {
        observed[t] = 0;
      } else {
        y[t] = rd.y();
      }
    }
    for (ba_engine *e : engines_) check(ba_ss_set_data(e, T, p, y.data(), X.data(), observed.data()));

    // ---- regression priors: BregVsSampler's ctor #5 pieces, kept and observed
    slab_ = slab;
    residual_precision_prior_ = residual_precision_prior;
    spike_ = spike;
    sigma_upper_limit_ = sigma_upper_limit;
    upload_regression_priors();
    observe();

    // ---- state models
    if (!structural_) {
      const LocalLevelStateModel *level = dynamic_cast<const LocalLevelStateModel *>(model->state_model(0));
      const DeviceStateVariancePrior &lp(variance_priors_[0]);
      for (ba_engine *e : engines_)
        check(ba_ss_set_local_level(e, prior_df(lp.precision_prior),
                                    prior_sigma_guess(lp.precision_prior), lp.sigma_upper_limit,
                                    level->initial_state_mean()[0],
                                    level->initial_state_variance()(0, 0),
                                    std::sqrt(level->sigsq())));
    } else {
      for (ba_engine *e : engines_) check(ba_ss_clear_state_models(e));
      size_t nseasonal = 0;
      for (int s = 0; s < nstate; ++s) {
        const Block &b(blocks_[s]);
        const StateModel *sm = model->state_model(s);
        double df[2] = {1, 1}, guess[2] = {1, 1}, upper[2] = {infinity(), infinity()}, init[2] = {1, 1};
        for (int v = 0; v < b.nvar; ++v) {
          const DeviceStateVariancePrior &pr(variance_priors_[b.var0 + v]);
          df[v] = prior_df(pr.precision_prior);
          guess[v] = prior_sigma_guess(pr.precision_prior);
          upper[v] = pr.sigma_upper_limit;
        }
        const Vector a0 = sm->initial_state_mean();
        const SpdMatrix V0 = sm->initial_state_variance();
        diagonal_or_die(V0, "a state model");
        const Vector v0 = V0.diag();
        int32_t ip[3] = {0, 1, 0};
        Vector phi;
        if (b.kind == 1) {
          init[0] = std::sqrt(dynamic_cast<const LocalLevelStateModel *>(sm)->sigsq());
        } else if (b.kind == 2) {
          const SpdMatrix Sigma = dynamic_cast<const LocalLinearTrendStateModel *>(sm)->Sigma();
          if (Sigma(0, 1) != 0.0)
            report_error("The device sampler draws the trend's two variances independently "
                         "(ZeroMeanMvnIndependenceSampler): Sigma must be diagonal.");
          init[0] = std::sqrt(Sigma(0, 0));
          init[1] = std::sqrt(Sigma(1, 1));
        } else if (b.kind == 3) {
          const SeasonalStateModel *seas = dynamic_cast<const SeasonalStateModel *>(sm);
          init[0] = std::sqrt(seas->sigsq());
          ip[0] = seas->nseasons();
          ip[1] = seas->season_duration();
          ip[2] = nseasonal < seasonal_time_of_first_observation.size()
                      ? seasonal_time_of_first_observation[nseasonal] : 0;
          ++nseasonal;
        } else if (b.kind == 4) {
          const ArStateModel *ar = dynamic_cast<const ArStateModel *>(sm);
          init[0] = ar->sigma();
          ip[0] = b.lags;
          phi = ar->phi();
        } else if (b.kind == 6) {
          // the rotations as the model's own transition matrix holds them (TrigStateModel.cpp:144-153:
          // [[cos, sin], [-sin, cos]] per frequency): no second cosine routine is involved
          TrigStateModel *trig = const_cast<TrigStateModel *>(dynamic_cast<const TrigStateModel *>(sm));
          init[0] = trig->error_distribution()->sigma();
          ip[0] = b.dim / 2;
          const Matrix Tm = sm->state_transition_matrix(0)->dense();
          phi.resize(b.dim);
          for (int q = 0; q < b.dim / 2; ++q) {
            phi[2 * q] = Tm(2 * q, 2 * q);
            phi[2 * q + 1] = Tm(2 * q, 2 * q + 1);
          }
        } else if (b.kind == 7) {
          const SemilocalLinearTrendStateModel *trend = dynamic_cast<const SemilocalLinearTrendStateModel *>(sm);
          const DeviceStateVariancePrior &sp(variance_priors_[b.var0 + 1]);
          if (!sp.slope_mean_prior || !sp.slope_ar1_prior)
            report_error("The slope entry of a SemilocalLinearTrendStateModel's state_variance_priors needs "
                         "slope_mean_prior and slope_ar1_prior.");
          init[0] = trend->level_sd();
          init[1] = trend->slope_sd();
          ip[0] = sp.force_stationary ? 1 : 0;
          ip[1] = sp.force_ar1_positive ? 1 : 0;
          phi.resize(6);
          phi[0] = sp.slope_mean_prior->mu();
          phi[1] = std::sqrt(sp.slope_mean_prior->sigsq());
          phi[2] = sp.slope_ar1_prior->mu();
          phi[3] = std::sqrt(sp.slope_ar1_prior->sigsq());
          phi[4] = trend->slope_mean();
          phi[5] = trend->slope_ar_coefficient();
        }   // (5, StaticInterceptStateModel: no parameter)
        for (ba_engine *e : engines_)
          check(ba_ss_add_state_model(e, b.kind, ip, df, guess, upper, init,
                                      (b.kind == 4 || b.kind == 6 || b.kind == 7) ? phi.data() : nullptr, a0.data(),
                                      v0.data()));
      }
    }

    // ---- the chains start where the model stands
    const RegressionModel *reg = model->observation_model();
    const Selector &inc(reg->coef().inc());
    std::vector<uint8_t> gamma(p, 0);
    for (int j = 0; j < p; ++j) gamma[j] = inc[j] ? 1 : 0;
    const Vector beta = reg->Beta();
    for (ba_engine *e : engines_) {
      check(ba_set_state(e, -1, gamma.data(), beta.data(), reg->sigsq()));
      if (lookahead > 1) check(ba_ss_set_lookahead(e, lookahead));
    }
  }

  DeviceStateSpacePosteriorSampler::~DeviceStateSpacePosteriorSampler() {
    unobserve();
    if (group_) ba_group_destroy(group_);
    else if (engine_) ba_engine_destroy(engine_);
  }

  ba_engine *DeviceStateSpacePosteriorSampler::locate(int chain, int64_t *local) const {
    if (!group_) {
      *local = chain;
      return engine_;
    }
    int32_t ei = 0;
    if (ba_group_locate(group_, chain, &ei, local) != BA_OK) report_error(ba_group_last_error());
    return engines_[ei];
  }

  void DeviceStateSpacePosteriorSampler::check(int rc) const {
    if (rc != BA_OK) report_error(ba_last_error());
  }

  void DeviceStateSpacePosteriorSampler::set_device_seed(unsigned long seed) {
    device_seed_ = seed;
    for (ba_engine *e : engines_) check(ba_seed(e, seed));
  }

  void DeviceStateSpacePosteriorSampler::record_state_of_chains(const std::vector<int> &chains) {
    std::vector<int64_t> c(1, 0);   // (chain 0 backs the model: always)
    for (int v : chains)
      if (v != 0) c.push_back(v);
    // (a device list: every engine keeps the chains that are its own)
    for (size_t i = 0; i < engines_.size(); ++i) {
      std::vector<int64_t> own;
      for (int64_t v : c) {
        int64_t local = 0;
        if (locate(static_cast<int>(v), &local) == engines_[i]) own.push_back(local);
      }
      if (i == 0 || !own.empty())
        check(ba_ss_lookahead_chains(engines_[i], static_cast<int32_t>(own.size()), own.data()));
    }
  }

  void DeviceStateSpacePosteriorSampler::upload_regression_priors() {
    const Vector mu = slab_->mu();
    const SpdMatrix ominv = slab_->unscaled_precision();
    const Vector pi = spike_->prior_inclusion_probabilities();
    for (ba_engine *e : engines_) {
      check(ba_set_slab(e, mu.data(), ominv.data()));
      check(ba_set_spike(e, pi.data(), spike_->max_model_size()));
      check(ba_set_sigma_prior(e, prior_df(residual_precision_prior_),
                               prior_sigma_guess(residual_precision_prior_), sigma_upper_limit_));
    }
    priors_stale_ = false;
  }
  void DeviceStateSpacePosteriorSampler::observe() {
    auto watch = [this](Model *m) {
      for (Ptr<Params> &prm : m->parameter_vector()) {
        prm->add_observer(this, [this]() { this->priors_stale_ = true; });
      }
    };
    watch(slab_.get());
    watch(residual_precision_prior_.get());
    watch(spike_.get());
  }
  void DeviceStateSpacePosteriorSampler::unobserve() {
    auto unwatch = [this](Model *m) {
      for (Ptr<Params> &prm : m->parameter_vector()) prm->remove_observer(this);
    };
    if (!!slab_) unwatch(slab_.get());
    if (!!residual_precision_prior_) unwatch(residual_precision_prior_.get());
    if (!!spike_) unwatch(spike_.get());
  }

  void DeviceStateSpacePosteriorSampler::draw() {
    // (a regression prior changed since the last draw: the setters are mutators behind the
    // look-ahead -- the chains go back to the draw the caller has seen, then take the values)
    if (priors_stale_) upload_regression_priors();
    // (one round of every chain; with the look-ahead the round has usually run already and
    // this hands out its record)
    for (ba_engine *e : engines_) check(ba_ss_draw_next(e));   // (every device's rounds are out before any is read)
    pull_chain0();
  }

  void DeviceStateSpacePosteriorSampler::chain_state(int chain, Selector &inc, Vector &beta,
                                                     double &sigsq, Vector &state_variances,
                                                     Matrix &state) const {
    const int p = model_->xdim(), T = model_->time_dimension();
    std::vector<uint8_t> gamma(p, 0);
    beta.resize(p);
    int64_t lc = 0;
    ba_engine *eng = locate(chain, &lc);
    check(ba_get_state(eng, lc, gamma.data(), beta.data(), &sigsq));
    inc = Selector(p, false);
    for (int j = 0; j < p; ++j)
      if (gamma[j]) inc.add(j);
    // the engine's layout -- step t at [t m, (t + 1) m) -- is a column-major m x T Matrix
    state = Matrix(state_dim_, T);
    state_variances = Vector(variance_priors_.size(), 0.0);
    if (structural_) {
      check(ba_ss_get_state_draw(eng, lc, state.data()));
      for (size_t s = 0; s < blocks_.size(); ++s) {
        if (blocks_[s].nvar == 0) continue;   // (a StaticInterceptStateModel has no parameter)
        check(ba_ss_get_state_model(eng, lc, static_cast<int32_t>(s),
                                    &state_variances[blocks_[s].var0], nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr));
      }
    } else {
      check(ba_ss_get_state(eng, lc, state.data(), &state_variances[0], nullptr, nullptr));
    }
  }

  void DeviceStateSpacePosteriorSampler::chain_ar(int chain, Vector &phi, double &sigsq, int which) const {
    int seen = 0;
    for (size_t s = 0; s < blocks_.size(); ++s) {
      if (blocks_[s].kind != 4 || seen++ != which) continue;
      phi.resize(blocks_[s].lags);
      int64_t lc = 0;
      ba_engine *eng = locate(chain, &lc);
      check(ba_ss_get_state_model(eng, lc, static_cast<int32_t>(s), &sigsq, nullptr, nullptr,
                                  phi.data(), nullptr, nullptr, nullptr, nullptr));
      return;
    }
    report_error("The model has no such ArStateModel.");
  }

  void DeviceStateSpacePosteriorSampler::pull_chain0() {
    Selector inc(model_->xdim(), false);
    Vector beta, variances;
    Matrix state;
    double sigsq = 1.0;
    chain_state(0, inc, beta, sigsq, variances, state);
    RegressionModel *reg = model_->observation_model();
    reg->coef().set_inc(inc);
    reg->set_included_coefficients(inc.select(beta));
    reg->set_sigsq(sigsq);
    for (size_t s = 0; s < blocks_.size(); ++s) {
      const Block &b(blocks_[s]);
      StateModel *sm = model_->state_model(static_cast<int>(s));
      if (b.kind == 1) {
        dynamic_cast<LocalLevelStateModel *>(sm)->set_sigsq(variances[b.var0]);
      } else if (b.kind == 2) {
        SpdMatrix Sigma(2, 0.0);
        Sigma(0, 0) = variances[b.var0];
        Sigma(1, 1) = variances[b.var0 + 1];
        dynamic_cast<LocalLinearTrendStateModel *>(sm)->set_Sigma(Sigma);
      } else if (b.kind == 3) {
        dynamic_cast<SeasonalStateModel *>(sm)->set_sigsq(variances[b.var0]);
      } else if (b.kind == 5) {
        // (StaticInterceptStateModel: nothing but its state, installed below)
      } else if (b.kind == 6) {
        dynamic_cast<TrigStateModel *>(sm)->error_distribution()->set_sigsq(variances[b.var0]);
      } else if (b.kind == 7) {
        // (the model's public setters take standard deviations)
        double pm[2] = {0.0, 0.0};
        check(ba_ss_get_state_model(engine_, 0, static_cast<int32_t>(s), nullptr, nullptr, nullptr, pm, nullptr,
                                    nullptr, nullptr, nullptr));
        SemilocalLinearTrendStateModel *trend = dynamic_cast<SemilocalLinearTrendStateModel *>(sm);
        trend->set_level_sd(std::sqrt(variances[b.var0]));
        trend->set_slope_sd(std::sqrt(variances[b.var0 + 1]));
        trend->set_slope_ar_coefficient(pm[0]);
        trend->set_slope_mean(pm[1]);
      } else {
        Vector phi(b.lags, 0.0);
        double ar_sigsq = 1.0;
        check(ba_ss_get_state_model(engine_, 0, static_cast<int32_t>(s), &ar_sigsq, nullptr, nullptr,
                                    phi.data(), nullptr, nullptr, nullptr, nullptr));
        ArStateModel *arm = dynamic_cast<ArStateModel *>(sm);
        arm->set_phi(phi);
        arm->set_sigsq(ar_sigsq);
      }
    }
    // the model's state matrix: the only public way to install one is
    // permanently_set_state (StateSpaceModelBase.cpp:199-212), which also tells the
    // model not to impute the state itself -- the device does
    model_->permanently_set_state(state);
  }

  // The observation model's and the state models' log priors
  // (StateSpacePosteriorSampler.cpp:66-74) at chain 0's draw: BregVsSampler::logpri from
  // the engine, GenericGaussianVarianceSampler::log_prior for every state variance
  // (GenericGaussianVarianceSampler.cpp: prior->logp(1 / sigsq) - 2 log(sigsq)).
  double DeviceStateSpacePosteriorSampler::logpri() const {
    double ans = negative_infinity();
    check(ba_logpri(engine_, 0, &ans));
    Selector inc(model_->xdim(), false);
    Vector beta, v;
    Matrix state;
    double sigsq_obs = 1.0;
    chain_state(0, inc, beta, sigsq_obs, v, state);
    for (size_t s = 0; s < blocks_.size(); ++s) {
      const Block &b(blocks_[s]);
      if (b.kind == 4) {
        // ArPosteriorSampler::log_prior_density (ArPosteriorSampler.cpp:66-70): the
        // stationarity indicator and the variance prior
        Vector phi(b.lags, 0.0);
        double sigsq = 1.0;
        check(ba_ss_get_state_model(engine_, 0, static_cast<int32_t>(s), &sigsq, nullptr, nullptr,
                                    phi.data(), nullptr, nullptr, nullptr, nullptr));
        if (!ArModel::check_stationary(phi)) return negative_infinity();
      }
      if (b.kind == 7) {
        // NonzeroMeanAr1Sampler::logpri (NonzeroMeanAr1Sampler.cpp:57-62): the long-run mean's and the
        // AR(1) coefficient's priors (the two variance priors below)
        double pm[2] = {0.0, 0.0};
        check(ba_ss_get_state_model(engine_, 0, static_cast<int32_t>(s), nullptr, nullptr, nullptr, pm, nullptr,
                                    nullptr, nullptr, nullptr));
        const DeviceStateVariancePrior &sp(variance_priors_[b.var0 + 1]);
        ans += sp.slope_mean_prior->logp(pm[1]) + sp.slope_ar1_prior->logp(pm[0]);
      }
      for (int k = 0; k < b.nvar; ++k) {
        const double sigsq = v[b.var0 + k];
        ans += variance_priors_[b.var0 + k].precision_prior->logp(1.0 / sigsq) - 2 * std::log(sigsq);
      }
    }
    return ans;
  }

  Matrix DeviceStateSpacePosteriorSampler::simulate_forecast(const Matrix &newX) {
    if (newX.ncol() != model_->xdim()) report_error("newX does not match the model's predictors.");
    const int h = newX.nrow();
    std::vector<double> out(static_cast<size_t>(chains_) * h);
    const size_t per = static_cast<size_t>(chains_) / engines_.size();
    for (size_t i = 0; i < engines_.size(); ++i)
      check(ba_ss_forecast(engines_[i], h, newX.data(), out.data() + i * per * h));
    Matrix ans(chains_, h);
    for (int c = 0; c < chains_; ++c)
      for (int t = 0; t < h; ++t) ans(c, t) = out[static_cast<size_t>(c) * h + t];
    return ans;
  }

}  // namespace BOOM
