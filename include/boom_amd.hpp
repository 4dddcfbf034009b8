// boom_amd.hpp -- header-only C++ host side over the C-ABI (boom_amd.h),
// mirroring the slice of BOOM's class surface that sits on the hot path so
// that callers (and our C++ parity test) read like code written against BOOM:
//
//   RegressionModel, GlmCoefs-style accessors   Models/Glm/RegressionModel.hpp:256-420
//   MvnGivenScalarSigma                         Models/MvnGivenScalarSigma.hpp:57-109
//   ChisqModel                                  Models/ChisqModel.hpp:28-39
//   VariableSelectionPrior                      Models/Glm/VariableSelectionPrior.hpp:99-135
//   BregVsSampler (5 ctors, setters, draw)      Models/Glm/PosteriorSamplers/BregVsSampler.hpp:64-161
//   StateSpaceRegressionModel                   Models/StateSpace/StateSpaceRegressionModel.hpp:82-113
//   LocalLevelStateModel                        Models/StateSpace/StateModels/LocalLevelStateModel.hpp
//   StateSpacePosteriorSampler                  Models/StateSpace/PosteriorSamplers/StateSpacePosteriorSampler.hpp:26-38
//   Model::sample_posterior / set_method        Models/ModelTypes.hpp:93-99, Policies/PriorPolicy.cpp:26-30
//   report_error -> std::runtime_error          cpputil/report_error.cpp:30-32
//
// Differences that are the point of the exercise: a sampler owns `chains`
// independent chains on one MI355X; draw() advances all of them; chain 0 backs
// the model's classic single-chain accessors; Ptr<T> is std::shared_ptr<T>.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "boom_amd.h"

namespace boom_amd_api {

template <class T>
using Ptr = std::shared_ptr<T>;
typedef std::vector<double> Vector;
typedef unsigned int uint;

inline void report_error(const std::string &msg) { throw std::runtime_error(msg); }
inline double infinity() { return std::numeric_limits<double>::infinity(); }

// column-major dense matrix, BOOM::Matrix layout (LinAlg/Matrix.hpp:429)
class Matrix {
 public:
  Matrix() : nr_(0), nc_(0) {}
  Matrix(int nr, int nc, double x = 0.0) : nr_(nr), nc_(nc), v_((size_t)nr * nc, x) {}
  int nrow() const { return nr_; }
  int ncol() const { return nc_; }
  double &operator()(int i, int j) { return v_[(size_t)j * nr_ + i]; }
  double operator()(int i, int j) const { return v_[(size_t)j * nr_ + i]; }
  double *data() { return v_.data(); }
  const double *data() const { return v_.data(); }
 private:
  int nr_, nc_;
  std::vector<double> v_;
};
typedef Matrix SpdMatrix;

// inclusion indicators (LinAlg/Selector.hpp)
class Selector {
 public:
  explicit Selector(int n = 0, bool all = false) : inc_(n, all ? 1 : 0) {}
  int nvars_possible() const { return (int)inc_.size(); }
  int nvars() const { int k = 0; for (auto b : inc_) k += b; return k; }
  bool operator[](int i) const { return inc_[i] != 0; }
  void add(int i) { inc_[i] = 1; }
  void drop(int i) { inc_[i] = 0; }
  void drop_all() { for (auto &b : inc_) b = 0; }
  std::vector<uint8_t> &bytes() { return inc_; }
  const std::vector<uint8_t> &bytes() const { return inc_; }
 private:
  std::vector<uint8_t> inc_;
};

// PosteriorSampler (Models/PosteriorSamplers/PosteriorSampler.hpp:49-66): the
// virtual surface of the path -- draw(), logpri(), the seed of the sampler's
// private stream(s)
class PosteriorSampler {
 public:
  virtual ~PosteriorSampler() {}
  virtual void draw() = 0;
  virtual double logpri() const = 0;
  virtual void set_seed(unsigned long seed) = 0;
};

class Model {
 public:
  virtual ~Model() {}
  void set_method(const Ptr<PosteriorSampler> &s) { samplers_.push_back(s); }
  void clear_methods() { samplers_.clear(); }
  int number_of_sampling_methods() const { return (int)samplers_.size(); }
  // PriorPolicy::sample_posterior
  void sample_posterior() { for (auto &s : samplers_) s->draw(); }
  Ptr<PosteriorSampler> sampler(int i) const { return samplers_[i]; }
 private:
  std::vector<Ptr<PosteriorSampler>> samplers_;
};

// ---- priors ---------------------------------------------------------------
struct MvnGivenScalarSigma {
  MvnGivenScalarSigma(const Vector &mean, const SpdMatrix &ominv) : mu_(mean), ominv_(ominv) {}
  const Vector &mu() const { return mu_; }
  const SpdMatrix &unscaled_precision() const { return ominv_; }
  int dim() const { return (int)mu_.size(); }
  Vector mu_;
  SpdMatrix ominv_;
};
struct ChisqModel {  // ChisqModel(df, sigma) == GammaModel(df/2, df sigma^2/2)
  explicit ChisqModel(double df = 1.0, double sigma_estimate = 1.0) : df_(df), sigma_(sigma_estimate) {}
  double df() const { return df_; }
  double sigma() const { return sigma_; }
  double alpha() const { return df_ / 2; }
  double beta() const { return df_ * sigma_ * sigma_ / 2; }
  double df_, sigma_;
};
struct VariableSelectionPrior {
  explicit VariableSelectionPrior(const Vector &pi) : pi_(pi), max_model_size_(-1) {}
  VariableSelectionPrior(uint n, double p) : pi_(n, p), max_model_size_(-1) {}
  void set_max_model_size(int64_t m) { max_model_size_ = m; }
  int64_t max_model_size() const { return max_model_size_; }
  const Vector &prior_inclusion_probabilities() const { return pi_; }
  uint potential_nvars() const { return (uint)pi_.size(); }
  Vector pi_;
  int64_t max_model_size_;
};

// BregVsSampler.hpp:33-40
struct ZellnerPriorParameters {
  Vector prior_inclusion_probabilities;
  Vector prior_beta_guess;
  double prior_beta_guess_weight;
  SpdMatrix prior_beta_information;  // Omega^{-1}
  double prior_sigma_guess;
  double prior_sigma_guess_weight;
};

// ---- engine handle shared by a model and its sampler ------------------------
class Engine {
 public:
  Engine(int chains, uint64_t seed, int device) {
    ba_config cfg{device, chains, 0, seed, 0, 0};
    if (ba_engine_create(&cfg, &e_) != BA_OK) report_error(ba_last_error());
    chains_ = chains;
  }
  // several devices of one node behind the handle (ba_group_*): engine i on devices[i]
  // owns the global chains [i * chains_per_device, (i + 1) * chains_per_device); get()
  // is the engine of chain 0.  Only RegressionModel / BregVsSampler take a device list.
  Engine(int chains_per_device, uint64_t seed, const std::vector<int> &devices) {
    std::vector<int32_t> dev(devices.begin(), devices.end());
    if (ba_group_create(dev.data(), (int32_t)dev.size(), chains_per_device, seed, &g_) != BA_OK)
      report_error(ba_group_last_error());
    e_ = ba_group_engine(g_, 0);
    chains_ = chains_per_device * (int)dev.size();
  }
  ~Engine() { if (g_) ba_group_destroy(g_); else ba_engine_destroy(e_); }
  Engine(const Engine &) = delete;
  ba_engine *get() const { return e_; }
  ba_group *group() const { return g_; }
  int size() const { return g_ ? ba_group_size(g_) : 1; }
  ba_engine *get(int i) const { return g_ ? ba_group_engine(g_, i) : e_; }
  int chains() const { return chains_; }
  void check(int rc) const { if (rc != BA_OK) report_error(ba_last_error()); }
  void check_group(int rc) const { if (rc != BA_OK) report_error(ba_group_last_error()); }
  // f(engine) on every engine of the handle
  template <class F> void each(F f) const { for (int i = 0; i < size(); ++i) check(f(get(i))); }
 private:
  ba_engine *e_ = nullptr;
  ba_group *g_ = nullptr;
  int chains_ = 0;
};

// ---- RegressionModel ----------------------------------------------------------
class RegressionModel : public Model {
 public:
  // RegressionModel(X, y, start_at_mle = false): sufficient statistics are
  // built on the device (NeRegSuf(X, y))
  RegressionModel(const Matrix &X, const Vector &y, int chains = 1,
                  uint64_t seed = 8675309, int device = 0)
      : eng_(new Engine(chains, seed, device)), p_(X.ncol()), inc_(X.ncol(), true),
        beta_(X.ncol(), 0.0), sigsq_(1.0) {
    if (X.nrow() != (int)y.size()) report_error("Number of rows of X must match the length of y.");
    eng_->check(ba_build_suf_from_xy(eng_->get(), X.nrow(), X.ncol(), X.data(), y.data()));
  }
  // the same over a device list: rows of X are sharded over the devices, one f64-MFMA
  // syrk per device and ONE all-reduce give every engine the identical NeRegSuf
  // (ba_group_build_suf_from_xy); `chains_per_device` chains on each entry of `devices`
  RegressionModel(const Matrix &X, const Vector &y, int chains_per_device,
                  const std::vector<int> &devices, uint64_t seed = 8675309)
      : eng_(new Engine(chains_per_device, seed, devices)), p_(X.ncol()), inc_(X.ncol(), true),
        beta_(X.ncol(), 0.0), sigsq_(1.0) {
    if (X.nrow() != (int)y.size()) report_error("Number of rows of X must match the length of y.");
    eng_->check_group(ba_group_build_suf_from_xy(eng_->group(), X.nrow(), X.ncol(), X.data(), y.data()));
  }
  int xdim() const { return p_; }
  int nvars_possible() const { return p_; }
  // (a change made through the mutators below goes to every chain before the
  // sampler's next draw, as BOOM's sampler reads the model's current state on
  // every draw)
  const Selector &inc() const { return inc_; }               // coef().inc()
  void set_inc(const Selector &g) { inc_ = g; dirty_ = true; }  // coef().set_inc(g)
  void drop_all() { inc_.drop_all(); dirty_ = true; }        // coef().drop_all()
  void add(int i) { inc_.add(i); dirty_ = true; }            // coef().add(i)
  void drop(int i) { inc_.drop(i); dirty_ = true; }          // coef().drop(i)
  const Vector &Beta() const { return beta_; }
  void set_Beta(const Vector &b) { beta_ = b; dirty_ = true; }
  double sigsq() const { return sigsq_; }
  void set_sigsq(double s) { sigsq_ = s; dirty_ = true; }
  bool dirty() const { return dirty_; }
  const Ptr<Engine> &engine() const { return eng_; }
  // all chains (chains x p row-major gamma / beta, chains sigsq)
  void chain_states(std::vector<uint8_t> &gamma, Vector &beta, Vector &sigsq) const {
    const size_t C = eng_->chains(), per = C / eng_->size();
    gamma.resize(C * p_); beta.resize(C * p_); sigsq.resize(C);
    for (int i = 0; i < eng_->size(); ++i)   // global chain order
      eng_->check(ba_get_states(eng_->get(i), gamma.data() + i * per * p_, beta.data() + i * per * p_,
                                sigsq.data() + i * per));
  }
  // used by the sampler
  void push_state() {
    eng_->each([&](ba_engine *e) { return ba_set_state(e, -1, inc_.bytes().data(), beta_.data(), sigsq_); });
    dirty_ = false;
  }
  // (while ba_draw_next serves a look-ahead batch, ba_get_state is the draw being served)
  void pull_chain0() {
    eng_->check(ba_get_state(eng_->get(), 0, inc_.bytes().data(), beta_.data(), &sigsq_));
    dirty_ = false;
  }
 private:
  Ptr<Engine> eng_;
  int p_;
  Selector inc_;
  Vector beta_;
  double sigsq_;
  bool dirty_ = true;
};

// ---- BregVsSampler ---------------------------------------------------------------
class BregVsSampler : public PosteriorSampler {
 public:
  // ctor #1 (BregVsSampler.cpp:48-85)
  BregVsSampler(RegressionModel *model, double prior_nobs, double expected_rsq,
                double expected_model_size, bool first_term_is_intercept = true)
      : model_(model) {
    all([&](ba_engine *e) { return ba_set_priors_ctor1(e, prior_nobs, expected_rsq, expected_model_size,
                                                        first_term_is_intercept); });
    all([&](ba_engine *e) { return ba_set_lookahead(e, kDefaultLookahead); });
  }
  // ctor #2 (BregVsSampler.cpp:87-142)
  BregVsSampler(RegressionModel *model, double prior_sigma_nobs, double prior_sigma_guess,
                double prior_beta_nobs, double diagonal_shrinkage,
                double prior_inclusion_probability, bool force_intercept = true)
      : model_(model) {
    all([&](ba_engine *e) { return ba_set_priors_ctor2(e, prior_sigma_nobs, prior_sigma_guess, prior_beta_nobs,
                                                        diagonal_shrinkage, prior_inclusion_probability,
                                                        force_intercept); });
    all([&](ba_engine *e) { return ba_set_lookahead(e, kDefaultLookahead); });
  }
  // ctor #3 (BregVsSampler.cpp:144-160)
  BregVsSampler(RegressionModel *model, const Vector &prior_mean,
                const SpdMatrix &unscaled_prior_precision, double sigma_guess, double df,
                const Vector &prior_inclusion_probs)
      : model_(model) {
    set_raw(prior_mean, unscaled_prior_precision, df, sigma_guess, prior_inclusion_probs, -1);
  }
  // ctor #4 (BregVsSampler.cpp:162-178)
  BregVsSampler(RegressionModel *model, const ZellnerPriorParameters &prior)
      : model_(model) {
    if ((int)prior.prior_beta_guess.size() != model->xdim()) report_error("Slab dimension did not match model dimension.");
    if ((int)prior.prior_inclusion_probabilities.size() != model->xdim()) report_error("Spike dimension did not match model dimension.");
    set_raw(prior.prior_beta_guess, prior.prior_beta_information, prior.prior_sigma_guess_weight,
            prior.prior_sigma_guess, prior.prior_inclusion_probabilities, -1);
  }
  // ctor #5 (BregVsSampler.cpp:180-194)
  BregVsSampler(RegressionModel *model, const Ptr<MvnGivenScalarSigma> &slab,
                const Ptr<ChisqModel> &residual_precision_prior,
                const Ptr<VariableSelectionPrior> &spike)
      : model_(model) {
    if (slab->dim() != model->xdim()) report_error("Slab dimension did not match model dimension.");
    if ((int)spike->potential_nvars() != model->xdim()) report_error("Spike dimension did not match model dimension.");
    set_raw(slab->mu(), slab->unscaled_precision(), residual_precision_prior->df(),
            residual_precision_prior->sigma(), spike->prior_inclusion_probabilities(),
            spike->max_model_size());
  }

  void draw() override {                     // BregVsSampler.cpp:252-261
    // the model's parameters as the caller left them (a host-side change since
    // the last draw goes to every chain first; the engine then rewinds any
    // look-ahead draws not handed out yet)
    if (model_->dirty()) model_->push_state();
    all([](ba_engine *e) { return ba_draw_next(e); });   // one launch per `lookahead` calls and device
    model_->pull_chain0();
  }
  // run `n` sweeps ahead per launch and hand them out one draw() at a time, for
  // EVERY chain (ba_set_lookahead: the draws are the ones one launch per call
  // would give, whatever is called in between)
  void set_lookahead(int n) { all([&](ba_engine *e) { return ba_set_lookahead(e, n < 1 ? 1 : n); }); }
  void draw(int nsweeps) {                   // many sweeps in one launch (per device, side by side)
    if (model_->dirty()) model_->push_state();
    all([&](ba_engine *e) { return ba_sweep(e, nsweeps); });
    model_->pull_chain0();
  }
  void limit_model_selection(uint n) { max_flips_ = (int)n; options(); }
  void suppress_model_selection() { max_flips_ = 0; options(); }
  void allow_model_selection(bool allow = true) { max_flips_ = allow ? -1 : 0; options(); }
  void suppress_beta_draw() { draw_beta_ = 0; options(); }
  void suppress_sigma_draw() { draw_sigma_ = 0; options(); }
  void set_correlation_swap_threshold(double t) { swap_ = t; options(); }
  void set_sigma_upper_limit(double s) {
    double df, ss;
    check(ba_get_priors(h(), nullptr, nullptr, nullptr, &df, &ss));
    all([&](ba_engine *e) { return ba_set_sigma_prior(e, df, std::sqrt(ss / df), s); });
  }
  void set_seed(unsigned long s) override { all([&](ba_engine *e) { return ba_seed(e, s); }); }
  double logpri() const override {           // BregVsSampler.cpp:380-393, chain 0 (the draw being served)
    double out;
    check(ba_logpri(h(), 0, &out));
    return out;
  }
  double prior_df() const { double df; check(ba_get_priors(h(), nullptr, nullptr, nullptr, &df, nullptr)); return df; }
  double prior_ss() const { double ss; check(ba_get_priors(h(), nullptr, nullptr, nullptr, nullptr, &ss)); return ss; }
  double log_model_prob(const Selector &g) const {
    double out;
    check(ba_log_model_prob(h(), 1, g.bytes().data(), &out));
    return out;
  }
 private:
  ba_engine *h() const { return model_->engine()->get(); }
  void check(int rc) const { model_->engine()->check(rc); }
  template <class F> void all(F f) const { model_->engine()->each(f); }
  void options() { all([&](ba_engine *e) { return ba_set_options(e, max_flips_, swap_, draw_beta_, draw_sigma_); }); }
  void set_raw(const Vector &b, const SpdMatrix &om, double df, double guess, const Vector &pi, int64_t mms) {
    all([&](ba_engine *e) { return ba_set_slab(e, b.data(), om.data()); });
    all([&](ba_engine *e) { return ba_set_spike(e, pi.data(), mms); });
    all([&](ba_engine *e) { return ba_set_sigma_prior(e, df, guess, infinity()); });
    // the caller's `for i: sample_posterior()` loop runs at the long-launch rate
    // by default; the draws do not depend on this number
    all([&](ba_engine *e) { return ba_set_lookahead(e, kDefaultLookahead); });
  }
  static constexpr int kDefaultLookahead = 256;
  RegressionModel *model_;
  int max_flips_ = -1, draw_beta_ = 1, draw_sigma_ = 1;
  double swap_ = 0.8;
};

// ---- bsts: local level + regression ----------------------------------------------
class LocalLevelStateModel {
 public:
  explicit LocalLevelStateModel(double sigma = 1.0) : sigma_(sigma) {}
  void set_initial_state_mean(double m) { a0_ = m; }
  void set_initial_state_variance(double v) { P0_ = v; }
  // ZeroMeanGaussianConjSampler(model, df, sigma_guess) + set_sigma_upper_limit
  void set_prior(double df, double sigma_guess, double sigma_upper_limit = infinity()) {
    df_ = df; guess_ = sigma_guess; upper_ = sigma_upper_limit;
  }
  double sigma_, a0_ = 0.0, P0_ = 1.0, df_ = 1.0, guess_ = 1.0, upper_ = infinity();
};

// LocalLinearTrendStateModel with one ZeroMeanMvnIndependenceSampler per variance
// (StateModels/LocalLinearTrend.cpp; the way bsts' AddLocalLinearTrend builds it)
class LocalLinearTrendStateModel {
 public:
  LocalLinearTrendStateModel() {}
  void set_initial_state_mean(const Vector &m) { a0_ = m; }
  void set_initial_state_variance(const Vector &diagonal) { P0_ = diagonal; }
  void set_initial_sigma(double level_sigma, double slope_sigma) { sigma_[0] = level_sigma; sigma_[1] = slope_sigma; }
  // ZeroMeanMvnIndependenceSampler(model, df, sigma_guess, which_variable) + set_sigma_upper_limit
  void set_prior(int which_variable, double df, double sigma_guess, double sigma_upper_limit = infinity()) {
    df_[which_variable] = df; guess_[which_variable] = sigma_guess; upper_[which_variable] = sigma_upper_limit;
  }
  Vector a0_ = Vector(2, 0.0), P0_ = Vector(2, 1.0);
  double sigma_[2] = {1.0, 1.0}, df_[2] = {1.0, 1.0}, guess_[2] = {1.0, 1.0}, upper_[2] = {infinity(), infinity()};
};
// SeasonalStateModel(nseasons, season_duration) with a ZeroMeanGaussianConjSampler: the
// transition is the seasonal matrix on the steps INTO a new season and the identity
// inside one (StateModels/SeasonalStateModel.cpp:89-104, :248-258)
class SeasonalStateModel {
 public:
  explicit SeasonalStateModel(int nseasons, int season_duration = 1)
      : nseasons_(nseasons), duration_(season_duration) {
    if (nseasons <= 0) report_error("'nseasons' must be positive in constructor for SeasonalStateModelBase");
    if (season_duration < 1) report_error("season_duration must be positive");
    a0_ = Vector(nseasons - 1, 0.0);
    P0_ = Vector(nseasons - 1, 1.0);
  }
  int state_dimension() const { return nseasons_ - 1; }
  int nseasons() const { return nseasons_; }
  int season_duration() const { return duration_; }
  void set_time_of_first_observation(int t0) { t0_ = t0; }
  void set_sigsq(double s) { sigma_ = std::sqrt(s); }
  void set_initial_state_mean(const Vector &m) { a0_ = m; }
  void set_initial_state_variance(double v) { P0_ = Vector(nseasons_ - 1, v); }
  void set_prior(double df, double sigma_guess, double sigma_upper_limit = infinity()) {
    df_ = df; guess_ = sigma_guess; upper_ = sigma_upper_limit;
  }
  int nseasons_, duration_, t0_ = 0;
  Vector a0_, P0_;
  double sigma_ = 1.0, df_ = 1.0, guess_ = 1.0, upper_ = infinity();
};

// ArStateModel(number_of_lags) with an ArPosteriorSampler (StateModels/ArStateModel.hpp:53,
// TimeSeries/PosteriorSamplers/ArPosteriorSampler.hpp)
class ArStateModel {
 public:
  explicit ArStateModel(int number_of_lags) : lags_(number_of_lags) {
    if (number_of_lags < 1) report_error("the number of lags must be positive in the constructor for ArStateModel");
    phi_ = Vector(number_of_lags, 0.0);
    a0_ = Vector(number_of_lags, 0.0);
    P0_ = Vector(number_of_lags, 1.0);
  }
  int state_dimension() const { return lags_; }
  int number_of_lags() const { return lags_; }
  void set_phi(const Vector &phi) { phi_ = phi; }
  void set_sigma(double s) { sigma_ = s; }
  void set_sigsq(double s) { sigma_ = std::sqrt(s); }
  void set_initial_state_mean(const Vector &m) { a0_ = m; }
  void set_initial_state_variance(double v) { P0_ = Vector(lags_, v); }
  // ArPosteriorSampler(model, ChisqModel(df, sigma_guess)) [+ set_sigma_upper_limit]
  void set_prior(double df, double sigma_guess, double sigma_upper_limit = infinity()) {
    df_ = df; guess_ = sigma_guess; upper_ = sigma_upper_limit;
  }
  int lags_;
  Vector phi_, a0_, P0_;
  double sigma_ = 1.0, df_ = 1.0, guess_ = 1.0, upper_ = infinity();
};

// StaticInterceptStateModel (StateModels/StaticInterceptStateModel.hpp:35-131): one component,
// T = 1, no state error, nothing to learn; set_initial_state_mean / _variance are its prior
class StaticInterceptStateModel {
 public:
  StaticInterceptStateModel() {}
  int state_dimension() const { return 1; }
  void set_initial_state_mean(double m) { a0_ = m; }
  void set_initial_state_variance(double v) {
    if (v < 0) report_error("Initial state variance must be non-negative.");
    P0_ = v;
  }
  double a0_ = 0.0, P0_ = 1.0;
};

// TrigStateModel(period, frequencies) (StateModels/TrigStateModel.hpp:83, .cpp:130-223): a
// cosine / sine pair per frequency that rotates by 2 pi f / period a step, one error variance
// for all components with a ZeroMeanGaussianConjSampler (set_prior), as bsts builds it
class TrigStateModel {
 public:
  TrigStateModel(double period, const Vector &frequencies) : period_(period), frequencies_(frequencies) {
    if (frequencies.size() == 0) report_error("At least one frequency needed to initialize TrigStateModel.");
    const int n = 2 * (int)frequencies.size();
    a0_ = Vector(n, 0.0);
    P0_ = Vector(n, 1.0);
    rotations_ = Vector(n, 0.0);
    for (int i = 0; i < (int)frequencies.size(); ++i) {
      const double freq = 2 * 3.141592653589793 * frequencies[i] / period;   // (Constants::pi)
      rotations_[2 * i] = std::cos(freq);
      rotations_[2 * i + 1] = std::sin(freq);
    }
  }
  int state_dimension() const { return (int)a0_.size(); }
  void set_sigsq(double s) { sigma_ = std::sqrt(s); }
  void set_initial_state_mean(const Vector &m) {
    if ((int)m.size() != state_dimension()) report_error("Argument to TrigStateModel::set_initial_state_mean is of the wrong size.");
    a0_ = m;
  }
  void set_initial_state_variance(const Vector &diagonal) {
    if ((int)diagonal.size() != state_dimension()) report_error("Argument to TrigStateModel::set_initial_state_variance has the wrong number of rows.");
    P0_ = diagonal;
  }
  // ZeroMeanGaussianConjSampler(error_distribution(), ChisqModel(df, sigma_guess)) + set_sigma_upper_limit
  void set_prior(double df, double sigma_guess, double sigma_upper_limit = infinity()) {
    df_ = df; guess_ = sigma_guess; upper_ = sigma_upper_limit;
  }
  double period_;
  Vector frequencies_, rotations_, a0_, P0_;
  double sigma_ = 1.0, df_ = 1.0, guess_ = 1.0, upper_ = infinity();
};

// SemilocalLinearTrendStateModel(level, slope) (StateModels/SemilocalLinearTrend.hpp:75): a level
// that moves by a slope which is itself a stationary AR(1) around a long-run mean -- state (level,
// slope, mean).  The two model objects it is built from (Models/ZeroMeanGaussianModel.hpp,
// Models/TimeSeries/NonzeroMeanAr1Model.hpp:77-84) hold the initial parameter values only.
class ZeroMeanGaussianModel {
 public:
  explicit ZeroMeanGaussianModel(double sigma = 1.0) : sigma_(sigma) {}
  double sigma() const { return sigma_; }
  double sigma_;
};
class NonzeroMeanAr1Model {
 public:
  explicit NonzeroMeanAr1Model(double mu = 0.0, double phi = 0.0, double sigma = 1.0) : mu_(mu), phi_(phi), sigma_(sigma) {}
  double mu() const { return mu_; }
  double phi() const { return phi_; }
  double sigma() const { return sigma_; }
  double mu_, phi_, sigma_;
};
class SemilocalLinearTrendStateModel {
 public:
  SemilocalLinearTrendStateModel(const Ptr<ZeroMeanGaussianModel> &level, const Ptr<NonzeroMeanAr1Model> &slope)
      : level_(level), slope_(slope) {}
  int state_dimension() const { return 3; }
  void set_initial_level_mean(double m) { a0_[0] = m; }
  void set_initial_level_sd(double sd) { P0_[0] = sd * sd; }
  void set_initial_slope_mean(double m) { a0_[1] = m; }
  void set_initial_slope_sd(double sd) { P0_[1] = sd * sd; }
  // ZeroMeanGaussianConjSampler(level, df, sigma_guess) + set_sigma_upper_limit
  void set_level_prior(double df, double sigma_guess, double sigma_upper_limit = infinity()) {
    df_[0] = df; guess_[0] = sigma_guess; upper_[0] = sigma_upper_limit;
  }
  // NonzeroMeanAr1Sampler(slope, GaussianModel(mean_mu, mean_sigma), GaussianModel(ar1_mu, ar1_sigma),
  // ChisqModel(df, sigma_guess)) + set_sigma_upper_limit, force_stationary(), force_ar1_positive()
  void set_slope_prior(double mean_mu, double mean_sigma, double ar1_mu, double ar1_sigma, double df,
                       double sigma_guess, double sigma_upper_limit = infinity(), bool force_stationary = true,
                       bool force_ar1_positive = false) {
    prior_[0] = mean_mu; prior_[1] = mean_sigma; prior_[2] = ar1_mu; prior_[3] = ar1_sigma;
    df_[1] = df; guess_[1] = sigma_guess; upper_[1] = sigma_upper_limit;
    force_stationary_ = force_stationary; force_positive_ = force_ar1_positive;
  }
  Ptr<ZeroMeanGaussianModel> level_;
  Ptr<NonzeroMeanAr1Model> slope_;
  double a0_[3] = {0.0, 0.0, 0.0}, P0_[3] = {1.0, 1.0, 0.0};
  double df_[2] = {1.0, 1.0}, guess_[2] = {1.0, 1.0}, upper_[2] = {infinity(), infinity()};
  double prior_[4] = {0.0, 1.0, 0.0, 1.0};
  bool force_stationary_ = true, force_positive_ = false;
};

class StateSpaceRegressionModel : public Model {
 public:
  StateSpaceRegressionModel(const Vector &y, const Matrix &X, const std::vector<bool> &observed,
                            int chains = 1, uint64_t seed = 8675309, int device = 0)
      : eng_(new Engine(chains, seed, device)), T_((int)y.size()), p_(X.ncol()) {
    if (X.nrow() != T_) report_error("X and y are incompatible in constructor for StateSpaceRegressionModel.");
    std::vector<uint8_t> obs;
    if (!observed.empty()) { obs.resize(T_); for (int t = 0; t < T_; ++t) obs[t] = observed[t]; }
    eng_->check(ba_ss_set_data(eng_->get(), T_, p_, y.data(), X.data(), obs.empty() ? nullptr : obs.data()));
  }
  // add_state, in any order and number (StateSpaceModelBase.hpp:637-638): the list goes to
  // the device when the sampler is attached or at the first draw (finalize_state)
  void add_state(const Ptr<LocalLevelStateModel> &s) { Entry e; e.kind = 1; e.level = s; models_.push_back(e); finalized_ = false; }
  void add_state(const Ptr<LocalLinearTrendStateModel> &s) { Entry e; e.kind = 2; e.trend = s; models_.push_back(e); finalized_ = false; }
  void add_state(const Ptr<SeasonalStateModel> &s) { Entry e; e.kind = 3; e.seasonal = s; models_.push_back(e); finalized_ = false; }
  void add_state(const Ptr<ArStateModel> &s) { Entry e; e.kind = 4; e.ar = s; models_.push_back(e); finalized_ = false; }
  void add_state(const Ptr<StaticInterceptStateModel> &s) { Entry e; e.kind = 5; e.intercept = s; models_.push_back(e); finalized_ = false; }
  void add_state(const Ptr<TrigStateModel> &s) { Entry e; e.kind = 6; e.trig = s; models_.push_back(e); finalized_ = false; }
  void add_state(const Ptr<SemilocalLinearTrendStateModel> &s) { Entry e; e.kind = 7; e.semilocal = s; models_.push_back(e); finalized_ = false; }
  int number_of_state_models() const { return (int)models_.size(); }
  // anything but a lone local level (which runs the local-level kernels)
  bool structural() const { return !(models_.size() == 1 && models_[0].kind == 1); }
  int state_dimension() const {
    int m = 0;
    for (const Entry &e : models_)
      m += (e.kind == 1 || e.kind == 5) ? 1 : e.kind == 2 ? 2 : e.kind == 3 ? e.seasonal->state_dimension()
           : e.kind == 6 ? e.trig->state_dimension() : e.kind == 7 ? 3 : e.ar->lags_;
    return m;
  }
  void finalize_state() {
    if (finalized_) return;
    if (models_.empty()) report_error("No state has been defined.");
    ba_engine *h = eng_->get();
    if (!structural()) {
      const LocalLevelStateModel &s = *models_[0].level;
      eng_->check(ba_ss_set_local_level(h, s.df_, s.guess_, s.upper_, s.a0_, s.P0_, s.sigma_));
    } else {
      eng_->check(ba_ss_clear_state_models(h));
      for (const Entry &e : models_) {
        if (e.kind == 1) {
          const LocalLevelStateModel &s = *e.level;
          eng_->check(ba_ss_add_state_model(h, 1, nullptr, &s.df_, &s.guess_, &s.upper_, &s.sigma_, nullptr, &s.a0_, &s.P0_));
        } else if (e.kind == 2) {
          const LocalLinearTrendStateModel &s = *e.trend;
          eng_->check(ba_ss_add_state_model(h, 2, nullptr, s.df_, s.guess_, s.upper_, s.sigma_, nullptr, s.a0_.data(), s.P0_.data()));
        } else if (e.kind == 3) {
          const SeasonalStateModel &s = *e.seasonal;
          const int32_t ip[3] = {s.nseasons_, s.duration_, s.t0_};
          eng_->check(ba_ss_add_state_model(h, 3, ip, &s.df_, &s.guess_, &s.upper_, &s.sigma_, nullptr, s.a0_.data(), s.P0_.data()));
        } else if (e.kind == 5) {
          const StaticInterceptStateModel &s = *e.intercept;
          eng_->check(ba_ss_add_state_model(h, 5, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &s.a0_, &s.P0_));
        } else if (e.kind == 7) {
          const SemilocalLinearTrendStateModel &s = *e.semilocal;
          const int32_t ip[3] = {s.force_stationary_ ? 1 : 0, s.force_positive_ ? 1 : 0, 0};
          const double sig[2] = {s.level_->sigma_, s.slope_->sigma_};
          const double pp[6] = {s.prior_[0], s.prior_[1], s.prior_[2], s.prior_[3], s.slope_->mu_, s.slope_->phi_};
          eng_->check(ba_ss_add_state_model(h, 7, ip, s.df_, s.guess_, s.upper_, sig, pp, s.a0_, s.P0_));
        } else if (e.kind == 6) {
          const TrigStateModel &s = *e.trig;
          const int32_t ip[3] = {(int32_t)s.frequencies_.size(), 0, 0};
          eng_->check(ba_ss_add_state_model(h, 6, ip, &s.df_, &s.guess_, &s.upper_, &s.sigma_, s.rotations_.data(), s.a0_.data(), s.P0_.data()));
        } else {
          const ArStateModel &s = *e.ar;
          const int32_t ip[3] = {s.lags_, 0, 0};
          eng_->check(ba_ss_add_state_model(h, 4, ip, &s.df_, &s.guess_, &s.upper_, &s.sigma_, s.phi_.data(), s.a0_.data(), s.P0_.data()));
        }
      }
    }
    finalized_ = true;
  }
  // the coefficients and error variance of the `which`-th ArStateModel in one chain's current draw
  Vector ar_phi(int chain = 0, int which = 0) const {
    const int b = ar_block(which);
    Vector v(models_[b].ar->lags_);
    eng_->check(ba_ss_get_state_model(eng_->get(), chain, b, nullptr, nullptr, nullptr, v.data(), nullptr, nullptr, nullptr, nullptr));
    return v;
  }
  double ar_sigsq(int chain = 0, int which = 0) const {
    double s2;
    eng_->check(ba_ss_get_state_model(eng_->get(), chain, ar_block(which), &s2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    return s2;
  }
  // the slope model of the `which`-th SemilocalLinearTrendStateModel in one chain's current draw:
  // (AR(1) coefficient, long-run mean)
  Vector semilocal_slope(int chain = 0, int which = 0) const {
    int seen = 0;
    for (size_t b = 0; b < models_.size(); ++b) {
      if (models_[b].kind != 7 || seen++ != which) continue;
      Vector v(2);
      eng_->check(ba_ss_get_state_model(eng_->get(), chain, (int32_t)b, nullptr, nullptr, nullptr, v.data(), nullptr, nullptr, nullptr, nullptr));
      return v;
    }
    report_error("The model has no such SemilocalLinearTrendStateModel.");
    return Vector();
  }
  // one chain's state draw: state_dimension x time_dimension, the models' components in
  // the order the models were added
  Matrix structural_state(int chain = 0) const {
    const int m = state_dimension();
    Vector buf((size_t)T_ * m);
    eng_->check(ba_ss_get_state_draw(eng_->get(), chain, buf.data()));
    Matrix st(m, T_);
    for (int t = 0; t < T_; ++t) for (int i = 0; i < m; ++i) st(i, t) = buf[(size_t)t * m + i];
    return st;
  }
  // every variance parameter in state-model order (a local linear trend has two: level,
  // slope; an autoregression's is its error variance; a static intercept has none)
  Vector state_variances(int chain = 0) const {
    std::vector<double> out;
    for (size_t b = 0; b < models_.size(); ++b) {
      if (models_[b].kind == 5) continue;
      double v[2] = {0, 0};
      eng_->check(ba_ss_get_state_model(eng_->get(), chain, (int32_t)b, v, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
      out.push_back(v[0]);
      if (models_[b].kind == 2 || models_[b].kind == 7) out.push_back(v[1]);
    }
    Vector ans(out.size());
    for (size_t i = 0; i < out.size(); ++i) ans[i] = out[i];
    return ans;
  }
  int time_dimension() const { return T_; }
  int xdim() const { return p_; }
  const Ptr<Engine> &engine() const { return eng_; }
  const LocalLevelStateModel *level() const { return structural() ? nullptr : models_[0].level.get(); }
  Vector state(int chain = 0) const {
    Vector st(T_);
    eng_->check(ba_ss_get_state(eng_->get(), chain, st.data(), nullptr, nullptr, nullptr));
    return st;
  }
  double level_sigsq(int chain = 0) const {
    double v;
    eng_->check(ba_ss_get_state(eng_->get(), chain, nullptr, &v, nullptr, nullptr));
    return v;
  }
 private:
  struct Entry {
    int kind = 0;   // 1 local level, 2 local linear trend, 3 seasonal, 4 autoregression, 5 static intercept, 6 trig, 7 semilocal linear trend
    Ptr<LocalLevelStateModel> level;
    Ptr<LocalLinearTrendStateModel> trend;
    Ptr<SeasonalStateModel> seasonal;
    Ptr<ArStateModel> ar;
    Ptr<StaticInterceptStateModel> intercept;
    Ptr<TrigStateModel> trig;
    Ptr<SemilocalLinearTrendStateModel> semilocal;
  };
  int ar_block(int which) const {
    int seen = 0;
    for (size_t b = 0; b < models_.size(); ++b)
      if (models_[b].kind == 4 && seen++ == which) return (int)b;
    report_error("The model has no such ArStateModel.");
    return -1;
  }
  Ptr<Engine> eng_;
  int T_, p_;
  std::vector<Entry> models_;
  bool finalized_ = false;
};

// regression priors are set through the same three pieces as BregVsSampler
class StateSpacePosteriorSampler : public PosteriorSampler {
 public:
  StateSpacePosteriorSampler(StateSpaceRegressionModel *model, const Ptr<MvnGivenScalarSigma> &slab,
                             const Ptr<ChisqModel> &residual_precision_prior,
                             const Ptr<VariableSelectionPrior> &spike,
                             double sigma_upper_limit = infinity())
      : model_(model) {
    ba_engine *h = model->engine()->get();
    model->engine()->check(ba_set_slab(h, slab->mu().data(), slab->unscaled_precision().data()));
    model->engine()->check(ba_set_spike(h, spike->prior_inclusion_probabilities().data(), spike->max_model_size()));
    model->engine()->check(ba_set_sigma_prior(h, residual_precision_prior->df(), residual_precision_prior->sigma(), sigma_upper_limit));
    std::vector<uint8_t> g0(model->xdim(), 0);
    model->engine()->check(ba_set_state(h, -1, g0.data(), nullptr, 1.0));
    // the callers' per-iteration loop at the device's rate (ba_ss_draw_next serves the
    // rounds from batches enqueued ahead; nothing a caller does can observe it)
    model->engine()->check(ba_ss_set_lookahead(h, 64));
  }
  void set_lookahead(int rounds) { model_->engine()->check(ba_ss_set_lookahead(model_->engine()->get(), rounds)); }
  void draw() override {                     // StateSpacePosteriorSampler.cpp:42-64
    model_->finalize_state();
    model_->engine()->check(ba_ss_draw_next(model_->engine()->get()));
  }
  // StateSpacePosteriorSampler::logpri (StateSpacePosteriorSampler.cpp:66-74) sums
  // the observation model's and the state models' log priors; the regression
  // part is what the engine evaluates (chain 0)
  double logpri() const override {
    double out;
    model_->engine()->check(ba_logpri(model_->engine()->get(), 0, &out));
    // + LocalLevelStateModel's sampler: ZeroMeanGaussianConjSampler::logpri =
    // GenericGaussianVarianceSampler::log_prior(sigma^2_level)
    // (GenericGaussianVarianceSampler.cpp: Gamma(df/2, ss/2) density of the
    // precision and the Jacobian of the reciprocal)
    if (const LocalLevelStateModel *lv = model_->level()) {
      const double a = lv->df_ / 2, b = lv->df_ * lv->guess_ * lv->guess_ / 2;
      const double s2 = model_->level_sigsq(0), x = 1.0 / s2;
      out += a * std::log(b) - std::lgamma(a) + (a - 1.0) * std::log(x) - b * x - 2.0 * std::log(s2);
    }
    return out;
  }
  void set_seed(unsigned long s) override {
    model_->engine()->check(ba_seed(model_->engine()->get(), s));
  }
 private:
  StateSpaceRegressionModel *model_;
};

// ---- binomial probit / logit spike and slab -----------------------------------------
struct MvnModel {   // a slab whose precision does not scale with sigma^2 (MvnBase)
  MvnModel(const Vector &mean, const SpdMatrix &precision) : mu_(mean), siginv_(precision) {}
  const Vector &mu() const { return mu_; }
  const SpdMatrix &siginv() const { return siginv_; }
  int dim() const { return (int)mu_.size(); }
  Vector mu_;
  SpdMatrix siginv_;
};

// BinomialLogitModel / BinomialProbitModel(X, y, n): the coefficients live in
// coef() = (inc(), Beta()); many chains on the device, chain 0 backs the model
class BinomialRegressionModelBase : public Model {
 public:
  BinomialRegressionModelBase(const Matrix &X, const Vector &y, const Vector &n, bool logit,
                              int clt_threshold, int chains, uint64_t seed, int device)
      : eng_(new Engine(chains, seed, device)), p_(X.ncol()), logit_(logit), inc_(X.ncol(), true),
        beta_(X.ncol(), 0.0) {
    if (X.nrow() != (int)y.size() || y.size() != n.size())
      report_error("X, y and n are incompatible in the binomial regression model's constructor.");
    eng_->check(logit ? ba_logit_set_data(eng_->get(), X.nrow(), X.ncol(), X.data(), y.data(), n.data(), clt_threshold)
                      : ba_probit_set_data(eng_->get(), X.nrow(), X.ncol(), X.data(), y.data(), n.data(), clt_threshold));
  }
  int xdim() const { return p_; }
  bool logit() const { return logit_; }
  const Selector &inc() const { return inc_; }
  void drop_all() { inc_.drop_all(); dirty_ = true; }
  void add(int i) { inc_.add(i); dirty_ = true; }
  void drop(int i) { inc_.drop(i); dirty_ = true; }
  const Vector &Beta() const { return beta_; }
  void set_Beta(const Vector &b) { beta_ = b; dirty_ = true; }
  bool dirty() const { return dirty_; }
  const Ptr<Engine> &engine() const { return eng_; }
  void push_state() {
    eng_->check(ba_set_state(eng_->get(), -1, inc_.bytes().data(), beta_.data(), 1.0));
    dirty_ = false;
  }
  void pull_chain0() {
    eng_->check(ba_get_state(eng_->get(), 0, inc_.bytes().data(), beta_.data(), nullptr));
    dirty_ = false;
  }
  void chain_states(std::vector<uint8_t> &gamma, Vector &beta) const {
    const size_t C = eng_->chains();
    gamma.resize(C * p_); beta.resize(C * p_);
    eng_->check(ba_get_states(eng_->get(), gamma.data(), beta.data(), nullptr));
  }
 private:
  Ptr<Engine> eng_;
  int p_;
  bool logit_;
  Selector inc_;
  Vector beta_;
  bool dirty_ = true;
};
class BinomialLogitModel : public BinomialRegressionModelBase {
 public:
  BinomialLogitModel(const Matrix &X, const Vector &y, const Vector &n, int clt_threshold = 5,
                     int chains = 1, uint64_t seed = 8675309, int device = 0)
      : BinomialRegressionModelBase(X, y, n, true, clt_threshold, chains, seed, device) {}
};
class BinomialProbitModel : public BinomialRegressionModelBase {
 public:
  BinomialProbitModel(const Matrix &X, const Vector &y, const Vector &n, int clt_threshold = 5,
                      int chains = 1, uint64_t seed = 8675309, int device = 0)
      : BinomialRegressionModelBase(X, y, n, false, clt_threshold, chains, seed, device) {}
};

// BinomialLogitSpikeSlabSampler / BinomialProbitSpikeSlabSampler(model, slab, spike, clt_threshold)
class BinomialSpikeSlabSamplerBase : public PosteriorSampler {
 public:
  BinomialSpikeSlabSamplerBase(BinomialRegressionModelBase *model, const Ptr<MvnModel> &slab,
                               const Ptr<VariableSelectionPrior> &spike)
      : model_(model), slab_(slab) {
    if (slab->dim() != model->xdim()) report_error("Slab does not match model dimension.");
    if ((int)spike->potential_nvars() != model->xdim()) report_error("Spike does not match model dimension.");
    check(ba_sss_set_slab(h(), slab->mu().data(), slab->siginv().data(), 0, -1));
    check(ba_set_spike(h(), spike->prior_inclusion_probabilities().data(), spike->max_model_size()));
  }
  void draw() override {
    if (model_->dirty()) model_->push_state();
    check(model_->logit() ? ba_logit_sweep(h(), 1) : ba_probit_sweep(h(), 1));
    check(ba_sync(h()));
    model_->pull_chain0();
  }
  double logpri() const override { report_error("logpri() is not implemented for the binomial samplers"); return 0; }
  void set_seed(unsigned long s) override { check(ba_seed(h(), s)); }
  void limit_model_selection(int max_flips) {
    check(ba_sss_set_slab(h(), slab_->mu().data(), slab_->siginv().data(), 0, max_flips));
  }
 private:
  ba_engine *h() const { return model_->engine()->get(); }
  void check(int rc) const { model_->engine()->check(rc); }
  BinomialRegressionModelBase *model_;
  Ptr<MvnModel> slab_;
};
class BinomialLogitSpikeSlabSampler : public BinomialSpikeSlabSamplerBase {
 public:
  BinomialLogitSpikeSlabSampler(BinomialLogitModel *model, const Ptr<MvnModel> &slab,
                                const Ptr<VariableSelectionPrior> &spike)
      : BinomialSpikeSlabSamplerBase(model, slab, spike) {}
};
class BinomialProbitSpikeSlabSampler : public BinomialSpikeSlabSamplerBase {
 public:
  BinomialProbitSpikeSlabSampler(BinomialProbitModel *model, const Ptr<MvnModel> &slab,
                                 const Ptr<VariableSelectionPrior> &spike)
      : BinomialSpikeSlabSamplerBase(model, slab, spike) {}
};


// ---- Poisson regression spike and slab ---------------------------------------------------
// PoissonRegressionModel + PoissonRegressionSpikeSlabSampler (Models/Glm/
// PoissonRegressionModel.hpp, PosteriorSamplers/PoissonRegressionSpikeSlabSampler.cpp:55-108;
// the data augmentation of PoissonDataImputer.cpp:36-96).  The normal mixtures that
// approximate the negative log-gamma densities are DATA of the reference
// (create_poisson_mixture_approximation_table, NormalMixtureApproximationTable): the caller
// hands them over, as a BOOM-side binding reads them from BOOM's own table
// (bindings/boom/DevicePoissonRegressionSpikeSlabSampler.cpp).
struct NormalMixtureTable {
  std::vector<int64_t> counts;     // ascending; 1 and every positive count of the data below largest_index
  std::vector<int32_t> ncomp;      // components of each count's mixture (<= 32)
  Vector mu, sigma, weight;        // the mixtures one after the other
  int64_t largest_index = 0;       // from here on the Gaussian limit is used
};
class PoissonRegressionModel : public Model {
 public:
  PoissonRegressionModel(const Matrix &X, const Vector &y, const Vector &exposure, int chains = 1,
                         uint64_t seed = 8675309, int device = 0)
      : eng_(new Engine(chains, seed, device)), p_(X.ncol()), inc_(X.ncol(), true), beta_(X.ncol(), 0.0) {
    if (X.nrow() != (int)y.size() || y.size() != exposure.size())
      report_error("X, y and exposure are incompatible in the Poisson regression model's constructor.");
    eng_->check(ba_poisson_set_data(eng_->get(), X.nrow(), X.ncol(), X.data(), y.data(), exposure.data()));
  }
  void set_mixture_table(const NormalMixtureTable &t) {
    if (t.counts.size() != t.ncomp.size()) report_error("the mixture table's counts and ncomp differ in length");
    eng_->check(ba_poisson_set_mixtures(eng_->get(), (int32_t)t.counts.size(), t.counts.data(), t.ncomp.data(),
                                        t.mu.data(), t.sigma.data(), t.weight.data(), t.largest_index));
  }
  int xdim() const { return p_; }
  const Selector &inc() const { return inc_; }
  void drop_all() { inc_.drop_all(); dirty_ = true; }
  void add(int i) { inc_.add(i); dirty_ = true; }
  void drop(int i) { inc_.drop(i); dirty_ = true; }
  const Vector &Beta() const { return beta_; }
  void set_Beta(const Vector &b) { beta_ = b; dirty_ = true; }
  bool dirty() const { return dirty_; }
  const Ptr<Engine> &engine() const { return eng_; }
  void push_state() {
    eng_->check(ba_set_state(eng_->get(), -1, inc_.bytes().data(), beta_.data(), 1.0));
    dirty_ = false;
  }
  void pull_chain0() {
    eng_->check(ba_get_state(eng_->get(), 0, inc_.bytes().data(), beta_.data(), nullptr));
    dirty_ = false;
  }
  void chain_states(std::vector<uint8_t> &gamma, Vector &beta) const {
    const size_t C = eng_->chains();
    gamma.resize(C * p_); beta.resize(C * p_);
    eng_->check(ba_get_states(eng_->get(), gamma.data(), beta.data(), nullptr));
  }
 private:
  Ptr<Engine> eng_;
  int p_;
  Selector inc_;
  Vector beta_;
  bool dirty_ = true;
};
// PoissonRegressionSpikeSlabSampler(model, slab, spike, number_of_threads, seeding_rng)
class PoissonRegressionSpikeSlabSampler : public PosteriorSampler {
 public:
  PoissonRegressionSpikeSlabSampler(PoissonRegressionModel *model, const Ptr<MvnModel> &slab,
                                    const Ptr<VariableSelectionPrior> &spike)
      : model_(model), slab_(slab) {
    if (slab->dim() != model->xdim()) report_error("Slab does not match model dimension.");
    if ((int)spike->potential_nvars() != model->xdim()) report_error("Spike does not match model dimension.");
    check(ba_sss_set_slab(h(), slab->mu().data(), slab->siginv().data(), 0, -1));
    check(ba_set_spike(h(), spike->prior_inclusion_probabilities().data(), spike->max_model_size()));
  }
  void draw() override {
    if (model_->dirty()) model_->push_state();
    check(ba_poisson_sweep(h(), 1));
    check(ba_sync(h()));
    model_->pull_chain0();
  }
  double logpri() const override { report_error("logpri() is not implemented for the Poisson sampler"); return 0; }
  void set_seed(unsigned long s) override { check(ba_seed(h(), s)); }
  void limit_model_selection(int max_flips) {
    check(ba_sss_set_slab(h(), slab_->mu().data(), slab_->siginv().data(), 0, max_flips));
  }
 private:
  ba_engine *h() const { return model_->engine()->get(); }
  void check(int rc) const { model_->engine()->check(rc); }
  PoissonRegressionModel *model_;
  Ptr<MvnModel> slab_;
};

}  // namespace boom_amd_api
