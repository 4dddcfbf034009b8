// See DeviceStateSpacePosteriorSampler.hpp.  OUR code against the reference's headers.
#include "DeviceStateSpacePosteriorSampler.hpp"

#include <cmath>
#include <cstring>

#include "Models/StateSpace/StateModels/LocalLevelStateModel.hpp"
#include "Models/StateSpace/StateModels/LocalLinearTrend.hpp"
#include "Models/StateSpace/StateModels/SeasonalStateModel.hpp"
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
      int device, RNG &seeding_rng)
      : PosteriorSampler(seeding_rng),
        model_(model),
        engine_(nullptr),
        chains_(chains),
        trend_(0),
        nseasons_(0),
        state_dim_(0),
        structural_(false),
        variance_priors_(state_variance_priors) {
    const int p = model->xdim();
    const int T = model->time_dimension();
    if (slab->dim() != p) report_error("Slab dimension did not match model dimension.");
    if (static_cast<int>(spike->potential_nvars()) != p)
      report_error("Spike dimension did not match model dimension.");
    if (T <= 0) report_error("Add the data to the model before creating the device sampler.");

    // ---- which state models: a trend (local level | local linear trend), then an
    // optional seasonal component
    int nstate = model->number_of_state_models();
    // an ArStateModel may come last (bsts: AddAr after the trend / seasonal components)
    const ArStateModel *ar = nstate >= 2
        ? dynamic_cast<const ArStateModel *>(model->state_model(nstate - 1)) : nullptr;
    ar_index_ = ar ? nstate - 1 : -1;
    ar_lags_ = ar ? ar->number_of_lags() : 0;
    if (ar) --nstate;
    if (nstate < 1 || nstate > 2)
      report_error("The device sampler takes a trend state model, optionally followed by a seasonal "
                   "one and / or an autoregression.");
    const LocalLevelStateModel *level = dynamic_cast<const LocalLevelStateModel *>(model->state_model(0));
    const LocalLinearTrendStateModel *llt =
        dynamic_cast<const LocalLinearTrendStateModel *>(model->state_model(0));
    if (!level && !llt)
      report_error("The first state model must be a LocalLevelStateModel or a LocalLinearTrendStateModel.");
    trend_ = level ? 1 : 2;
    const SeasonalStateModel *seasonal = nullptr;
    if (nstate == 2) {
      seasonal = dynamic_cast<const SeasonalStateModel *>(model->state_model(1));
      if (!seasonal) report_error("The second state model must be a SeasonalStateModel.");
      if (seasonal->season_duration() != 1)
        report_error("Season durations other than 1 are not implemented on the device.");
      nseasons_ = seasonal->nseasons();
    }
    state_dim_ = trend_ + (nseasons_ > 0 ? nseasons_ - 1 : 0) + ar_lags_;
    structural_ = (trend_ == 2) || (nseasons_ > 0) || ar;
    const size_t nvar = static_cast<size_t>(trend_ + (nseasons_ > 0 ? 1 : 0) + (ar ? 1 : 0));
    if (variance_priors_.size() != nvar)
      report_error("state_variance_priors needs one entry per state variance parameter.");

    device_seed_ = seed_rng(seeding_rng);
    ba_config cfg{device, chains, 0, static_cast<uint64_t>(device_seed_), 0, 0};
    check(ba_engine_create(&cfg, &engine_));

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
    check(ba_ss_set_data(engine_, T, p, y.data(), X.data(), observed.data()));

    // ---- regression priors: BregVsSampler's ctor #5 pieces
    const Vector mu = slab->mu();
    const SpdMatrix ominv = slab->unscaled_precision();
    check(ba_set_slab(engine_, mu.data(), ominv.data()));
    const Vector pi = spike->prior_inclusion_probabilities();
    check(ba_set_spike(engine_, pi.data(), spike->max_model_size()));
    check(ba_set_sigma_prior(engine_, prior_df(residual_precision_prior),
                             prior_sigma_guess(residual_precision_prior), sigma_upper_limit));

    // ---- state models
    if (!structural_) {
      const DeviceStateVariancePrior &lp(variance_priors_[0]);
      check(ba_ss_set_local_level(engine_, prior_df(lp.precision_prior),
                                  prior_sigma_guess(lp.precision_prior), lp.sigma_upper_limit,
                                  level->initial_state_mean()[0],
                                  level->initial_state_variance()(0, 0),
                                  std::sqrt(level->sigsq())));
    } else {
      double df[3] = {1, 1, 1}, guess[3] = {1, 1, 1}, init[3] = {1, 1, 1};
      double upper[3] = {infinity(), infinity(), infinity()};
      Vector a0(state_dim_, 0.0), P0(state_dim_, 1.0);
      auto set_var = [&](int slot, const DeviceStateVariancePrior &pr, double sigsq) {
        df[slot] = prior_df(pr.precision_prior);
        guess[slot] = prior_sigma_guess(pr.precision_prior);
        upper[slot] = pr.sigma_upper_limit;
        init[slot] = std::sqrt(sigsq);
      };
      if (level) {
        set_var(0, variance_priors_[0], level->sigsq());
        a0[0] = level->initial_state_mean()[0];
        P0[0] = level->initial_state_variance()(0, 0);
      } else {
        const SpdMatrix Sigma = llt->Sigma();
        if (Sigma(0, 1) != 0.0)
          report_error("The device sampler draws the trend's two variances independently "
                       "(ZeroMeanMvnIndependenceSampler): Sigma must be diagonal.");
        set_var(0, variance_priors_[0], Sigma(0, 0));
        set_var(1, variance_priors_[1], Sigma(1, 1));
        const Vector m = llt->initial_state_mean();
        const SpdMatrix V = llt->initial_state_variance();
        diagonal_or_die(V, "the local linear trend");
        for (int i = 0; i < 2; ++i) {
          a0[i] = m[i];
          P0[i] = V(i, i);
        }
      }
      if (seasonal) {
        set_var(2, variance_priors_[trend_], seasonal->sigsq());
        const Vector m = seasonal->initial_state_mean();
        const SpdMatrix V = seasonal->initial_state_variance();
        diagonal_or_die(V, "the seasonal component");
        for (int i = 0; i < nseasons_ - 1; ++i) {
          a0[trend_ + i] = m[i];
          P0[trend_ + i] = V(i, i);
        }
      }
      check(ba_ss_set_structural(engine_, trend_, nseasons_, df, guess, upper, init, a0.data(),
                                 P0.data()));
      if (ar) {
        // ArPosteriorSampler(model, siginv_prior) + set_sigma_upper_limit
        const DeviceStateVariancePrior &pr(variance_priors_.back());
        const Vector m = ar->initial_state_mean();
        const SpdMatrix V = ar->initial_state_variance();
        diagonal_or_die(V, "the autoregression");
        const Vector v0 = V.diag();
        const Vector phi = ar->phi();
        check(ba_ss_add_ar(engine_, ar_lags_, prior_df(pr.precision_prior),
                           prior_sigma_guess(pr.precision_prior), pr.sigma_upper_limit, ar->sigma(),
                           phi.data(), m.data(), v0.data()));
      }
    }

    // ---- the chains start where the model stands
    const RegressionModel *reg = model->observation_model();
    const Selector &inc(reg->coef().inc());
    std::vector<uint8_t> gamma(p, 0);
    for (int j = 0; j < p; ++j) gamma[j] = inc[j] ? 1 : 0;
    const Vector beta = reg->Beta();
    check(ba_set_state(engine_, -1, gamma.data(), beta.data(), reg->sigsq()));
  }

  DeviceStateSpacePosteriorSampler::~DeviceStateSpacePosteriorSampler() {
    ba_engine_destroy(engine_);
  }

  void DeviceStateSpacePosteriorSampler::check(int rc) const {
    if (rc != BA_OK) report_error(ba_last_error());
  }

  void DeviceStateSpacePosteriorSampler::set_device_seed(unsigned long seed) {
    device_seed_ = seed;
    check(ba_seed(engine_, seed));
  }

  void DeviceStateSpacePosteriorSampler::draw() {
    check(ba_ss_sweep(engine_, 1));
    check(ba_sync(engine_));
    pull_chain0();
  }

  void DeviceStateSpacePosteriorSampler::chain_state(int chain, Selector &inc, Vector &beta,
                                                     double &sigsq, Vector &state_variances,
                                                     Matrix &state) const {
    const int p = model_->xdim(), T = model_->time_dimension();
    std::vector<uint8_t> gamma(p, 0);
    beta.resize(p);
    check(ba_get_state(engine_, chain, gamma.data(), beta.data(), &sigsq));
    inc = Selector(p, false);
    for (int j = 0; j < p; ++j)
      if (gamma[j]) inc.add(j);
    // the engine's layout -- step t at [t m, (t + 1) m) -- is a column-major m x T Matrix
    state = Matrix(state_dim_, T);
    state_variances = Vector(3, 0.0);
    if (structural_) {
      check(ba_ss_get_structural(engine_, chain, state.data(), state_variances.data(), nullptr,
                                 nullptr));
    } else {
      check(ba_ss_get_state(engine_, chain, state.data(), &state_variances[0], nullptr, nullptr));
    }
  }

  void DeviceStateSpacePosteriorSampler::chain_ar(int chain, Vector &phi, double &sigsq) const {
    if (ar_index_ < 0) report_error("The model has no ArStateModel.");
    phi.resize(ar_lags_);
    check(ba_ss_get_ar(engine_, chain, phi.data(), &sigsq, nullptr, nullptr, nullptr, nullptr));
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
    if (trend_ == 1) {
      dynamic_cast<LocalLevelStateModel *>(model_->state_model(0))->set_sigsq(variances[0]);
    } else {
      SpdMatrix Sigma(2, 0.0);
      Sigma(0, 0) = variances[0];
      Sigma(1, 1) = variances[1];
      dynamic_cast<LocalLinearTrendStateModel *>(model_->state_model(0))->set_Sigma(Sigma);
    }
    if (nseasons_ > 0)
      dynamic_cast<SeasonalStateModel *>(model_->state_model(1))->set_sigsq(variances[2]);
    if (ar_index_ >= 0) {
      Vector phi(ar_lags_, 0.0);
      double ar_sigsq = 1.0;
      check(ba_ss_get_ar(engine_, 0, phi.data(), &ar_sigsq, nullptr, nullptr, nullptr, nullptr));
      ArStateModel *arm = dynamic_cast<ArStateModel *>(model_->state_model(ar_index_));
      arm->set_phi(phi);
      arm->set_sigsq(ar_sigsq);
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
    Vector v(3, 0.0);
    if (structural_) {
      check(ba_ss_get_structural(engine_, 0, nullptr, v.data(), nullptr, nullptr));
    } else {
      check(ba_ss_get_state(engine_, 0, nullptr, &v[0], nullptr, nullptr));
    }
    const size_t nplain = variance_priors_.size() - (ar_index_ >= 0 ? 1 : 0);
    for (size_t i = 0; i < nplain; ++i) {
      const int slot = (i < static_cast<size_t>(trend_)) ? static_cast<int>(i) : 2;
      const double sigsq = v[slot];
      ans += variance_priors_[i].precision_prior->logp(1.0 / sigsq) - 2 * std::log(sigsq);
    }
    if (ar_index_ >= 0) {
      // ArPosteriorSampler::log_prior_density (ArPosteriorSampler.cpp:66-70): the
      // stationarity indicator and the variance prior
      Vector phi(ar_lags_, 0.0);
      double sigsq = 1.0;
      check(ba_ss_get_ar(engine_, 0, phi.data(), &sigsq, nullptr, nullptr, nullptr, nullptr));
      if (!ArModel::check_stationary(phi)) return negative_infinity();
      ans += variance_priors_.back().precision_prior->logp(1.0 / sigsq) - 2 * std::log(sigsq);
    }
    return ans;
  }

  Matrix DeviceStateSpacePosteriorSampler::simulate_forecast(const Matrix &newX) {
    if (newX.ncol() != model_->xdim()) report_error("newX does not match the model's predictors.");
    const int h = newX.nrow();
    std::vector<double> out(static_cast<size_t>(chains_) * h);
    check(ba_ss_forecast(engine_, h, newX.data(), out.data()));
    Matrix ans(chains_, h);
    for (int c = 0; c < chains_; ++c)
      for (int t = 0; t < h; ++t) ans(c, t) = out[static_cast<size_t>(c) * h + t];
    return ans;
  }

}  // namespace BOOM
