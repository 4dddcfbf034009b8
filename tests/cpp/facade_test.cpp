// C++ test of the BOOM-shaped host side (include/boom_amd.hpp) over the C-ABI.
// It re-expresses the reference's own acceptance tests for this path on
// many-chain output (Models/Glm/tests/regression_spike_slab_test.cc):
//   Small                :69-120   posterior covers the truth
//   TestMaxSizeControl   :124-171  set_max_model_size(2) => |gamma| <= 2
//   Large                :173-205  p = 100 => P(|gamma| <= 8) >= .95
//   PerfectCollinearity  :207-257  collinear columns share the inclusion
// plus the error conventions (report_error -> std::runtime_error) and the
// state-space classes.  Needs a GPU.  Prints "ALL OK" on success.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <string>

#include "boom_amd.hpp"

using namespace boom_amd_api;

static int failures = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);      \
      ++failures;                                                        \
    }                                                                    \
  } while (0)

struct Sim {
  Matrix X;
  Vector y, beta;
};

static Sim simulate(int n, int p, const Vector &beta, double sd, unsigned seed) {
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> N(0, 1);
  Sim s{Matrix(n, p), Vector(n), beta};
  for (int i = 0; i < n; ++i) {
    s.X(i, 0) = 1.0;
    for (int j = 1; j < p; ++j) s.X(i, j) = N(gen);
  }
  for (int i = 0; i < n; ++i) {
    double mu = 0;
    for (int j = 0; j < p; ++j) mu += s.X(i, j) * beta[j];
    s.y[i] = mu + sd * N(gen);
  }
  return s;
}

static void Small() {
  const int n = 1000, p = 10, chains = 64, niter = 200;
  Vector beta(p, 0.0);
  beta[0] = 1.5; beta[1] = -2.0; beta[2] = 3.0;
  Sim s = simulate(n, p, beta, 1.0, 1);
  RegressionModel model(s.X, s.y, chains, 8675309);
  Ptr<BregVsSampler> sampler(new BregVsSampler(&model, 1.0, 0.5, 3.0, true));
  model.set_method(sampler);
  model.drop_all();
  model.add(0);
  for (int i = 0; i < 50; ++i) model.sample_posterior();   // burn-in, one draw() per call
  Vector mean(p, 0.0), sig(1, 0.0);
  for (int i = 0; i < niter; ++i) {
    model.sample_posterior();
    std::vector<uint8_t> g; Vector b, s2;
    model.chain_states(g, b, s2);
    for (int c = 0; c < chains; ++c) {
      for (int j = 0; j < p; ++j) mean[j] += b[(size_t)c * p + j];
      sig[0] += std::sqrt(s2[c]);
    }
  }
  const double N = (double)niter * chains;
  for (int j = 0; j < p; ++j) EXPECT(std::fabs(mean[j] / N - beta[j]) < 0.12);
  EXPECT(std::fabs(sig[0] / N - 1.0) < 0.08);
  EXPECT(model.inc()[0] && model.sigsq() > 0);   // chain 0 through the classic accessors
}

static void TestMaxSizeControl() {
  const int n = 500, p = 12, chains = 32;
  Vector beta(p, 0.0);
  for (int j = 0; j < 6; ++j) beta[j] = 2.0 + j;
  Sim s = simulate(n, p, beta, 1.0, 2);
  RegressionModel model(s.X, s.y, chains, 11);
  Vector b(p, 0.0);
  SpdMatrix om(p, p, 0.0);
  for (int j = 0; j < p; ++j) om(j, j) = 0.01;
  Ptr<MvnGivenScalarSigma> slab(new MvnGivenScalarSigma(b, om));
  Ptr<ChisqModel> siginv(new ChisqModel(1.0, 1.0));
  Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(p, 0.5));
  spike->set_max_model_size(2);
  Ptr<BregVsSampler> sampler(new BregVsSampler(&model, slab, siginv, spike));
  model.set_method(sampler);
  model.drop_all();
  int worst = 0;
  for (int i = 0; i < 100; ++i) {
    model.sample_posterior();
    std::vector<uint8_t> g; Vector bb, s2;
    model.chain_states(g, bb, s2);
    for (int c = 0; c < chains; ++c) {
      int k = 0;
      for (int j = 0; j < p; ++j) k += g[(size_t)c * p + j];
      worst = std::max(worst, k);
    }
  }
  EXPECT(worst <= 2);
}

static void Large() {
  const int n = 2000, p = 100, chains = 64;
  Vector beta(p, 0.0);
  beta[0] = 1; beta[1] = 3; beta[2] = -3; beta[3] = 2;
  Sim s = simulate(n, p, beta, 1.0, 3);
  RegressionModel model(s.X, s.y, chains, 5);
  Ptr<BregVsSampler> sampler(new BregVsSampler(&model, 1.0, 0.5, 4.0, true));
  model.set_method(sampler);
  model.drop_all();
  model.add(0);
  sampler->draw(100);
  int small = 0, total = 0;
  for (int i = 0; i < 50; ++i) {
    sampler->draw(4);
    std::vector<uint8_t> g; Vector bb, s2;
    model.chain_states(g, bb, s2);
    for (int c = 0; c < chains; ++c) {
      int k = 0;
      for (int j = 0; j < p; ++j) k += g[(size_t)c * p + j];
      small += (k <= 8);
      ++total;
    }
  }
  EXPECT(small >= 0.95 * total);
}

static void PerfectCollinearity() {
  const int n = 400, p = 8, chains = 128;
  Vector beta(p, 0.0);
  beta[0] = 1.0; beta[1] = 2.0;
  Sim s = simulate(n, p, beta, 1.0, 4);
  // columns 2 and 3 are (near) copies of column 1
  std::mt19937_64 gen(99);
  std::normal_distribution<double> N(0, 1);
  for (int i = 0; i < n; ++i) {
    s.X(i, 2) = s.X(i, 1) + 1e-3 * N(gen);
    s.X(i, 3) = s.X(i, 1) + 1e-3 * N(gen);
  }
  RegressionModel model(s.X, s.y, chains, 17);
  Ptr<BregVsSampler> sampler(new BregVsSampler(&model, 1.0, 1.0, 0.5, 0.5, 0.3, true));
  model.set_method(sampler);
  model.drop_all();
  model.add(0);
  sampler->draw(200);
  double inc[4] = {0, 0, 0, 0};
  int draws = 0;
  for (int i = 0; i < 100; ++i) {
    sampler->draw(5);
    std::vector<uint8_t> g; Vector bb, s2;
    model.chain_states(g, bb, s2);
    for (int c = 0; c < chains; ++c) {
      for (int j = 1; j <= 3; ++j) inc[j] += g[(size_t)c * p + j];
      ++draws;
    }
  }
  // the signal is shared among the three copies: at least one is in, none
  // dominates completely
  const double any = (inc[1] + inc[2] + inc[3]) / draws;
  EXPECT(any > 0.95);
  for (int j = 1; j <= 3; ++j) EXPECT(inc[j] / draws > 0.1 && inc[j] / draws < 0.7);
}

static void ErrorConventions() {
  Vector beta(4, 0.0);
  Sim s = simulate(50, 4, beta, 1.0, 5);
  RegressionModel model(s.X, s.y, 2, 1);
  bool threw = false;
  try {
    Ptr<MvnGivenScalarSigma> slab(new MvnGivenScalarSigma(Vector(3, 0.0), SpdMatrix(3, 3, 0.0)));
    Ptr<ChisqModel> siginv(new ChisqModel(1.0, 1.0));
    Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(4, 0.5));
    BregVsSampler bad(&model, slab, siginv, spike);
  } catch (std::runtime_error &e) {
    threw = std::string(e.what()).find("Slab dimension") != std::string::npos;
  }
  EXPECT(threw);
  threw = false;
  try {
    BregVsSampler bad(&model, 1.0, 1.0, 0.5, 1.5 /* illegal shrinkage */, 0.3, true);
  } catch (std::runtime_error &e) {
    threw = std::string(e.what()).find("diagonal_shrinkage") != std::string::npos;
  }
  EXPECT(threw);
}

static void StateSpace() {
  const int T = 300, p = 4, chains = 16;
  std::mt19937_64 gen(7);
  std::normal_distribution<double> N(0, 1);
  Matrix X(T, p);
  Vector y(T), coef = {5.0, -4.0, 0.0, 0.0};
  double level = 0;
  for (int t = 0; t < T; ++t) {
    level += 0.3 * N(gen);
    double mu = level;
    for (int j = 0; j < p; ++j) { X(t, j) = N(gen); mu += X(t, j) * coef[j]; }
    y[t] = mu + 0.2 * N(gen);
  }
  StateSpaceRegressionModel model(y, X, std::vector<bool>(), chains, 3);
  Ptr<LocalLevelStateModel> level_model(new LocalLevelStateModel(1.0));
  level_model->set_initial_state_mean(y[0]);
  level_model->set_initial_state_variance(4.0);
  level_model->set_prior(1.0, 0.3);
  model.add_state(level_model);
  SpdMatrix om(p, p, 0.0);
  for (int j = 0; j < p; ++j) om(j, j) = 0.01;
  Ptr<MvnGivenScalarSigma> slab(new MvnGivenScalarSigma(Vector(p, 0.0), om));
  Ptr<ChisqModel> siginv(new ChisqModel(1.0, 0.5));
  Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(p, 0.5));
  Ptr<StateSpacePosteriorSampler> sampler(new StateSpacePosteriorSampler(&model, slab, siginv, spike));
  model.set_method(sampler);
  for (int i = 0; i < 100; ++i) model.sample_posterior();
  Vector st = model.state(3);
  EXPECT((int)st.size() == T);
  double err = 0;
  // the drawn state tracks y - X coef
  for (int t = 0; t < T; ++t) {
    double target = y[t];
    for (int j = 0; j < p; ++j) target -= X(t, j) * coef[j];
    err += std::fabs(st[t] - target);
  }
  EXPECT(err / T < 0.5);
  EXPECT(model.level_sigsq(3) > 0.01 && model.level_sigsq(3) < 1.0);
}

// bsts' standard model: regression + local linear trend + seasonal state
static void StructuralTimeSeries() {
  const int T = 420, p = 3, chains = 8, S = 7;
  std::mt19937_64 gen(11);
  std::normal_distribution<double> N(0, 1);
  Matrix X(T, p);
  Vector y(T), coef = {4.0, 0.0, -3.0};
  double pattern[S] = {1.5, -0.5, 0.3, -1.2, 0.8, -0.6, -0.3};
  double level = 0, slope = 0.03;
  for (int t = 0; t < T; ++t) {
    slope += 0.002 * N(gen);
    level += slope + 0.05 * N(gen);
    double mu = level + pattern[t % S];
    for (int j = 0; j < p; ++j) { X(t, j) = N(gen); mu += X(t, j) * coef[j]; }
    y[t] = mu + 0.2 * N(gen);
  }
  StateSpaceRegressionModel model(y, X, std::vector<bool>(), chains, 3);
  Ptr<LocalLinearTrendStateModel> trend(new LocalLinearTrendStateModel);
  trend->set_initial_state_mean(Vector{y[0], 0.0});
  trend->set_initial_state_variance(Vector{4.0, 1.0});
  trend->set_initial_sigma(0.5, 0.1);
  trend->set_prior(0, 1.0, 0.1);
  trend->set_prior(1, 1.0, 0.01);
  model.add_state(trend);
  Ptr<SeasonalStateModel> seasonal(new SeasonalStateModel(S));
  seasonal->set_sigsq(0.01);
  seasonal->set_initial_state_variance(4.0);
  seasonal->set_prior(1.0, 0.05);
  model.add_state(seasonal);
  EXPECT(model.state_dimension() == 2 + S - 1);
  SpdMatrix om(p, p, 0.0);
  for (int j = 0; j < p; ++j) om(j, j) = 0.01;
  Ptr<MvnGivenScalarSigma> slab(new MvnGivenScalarSigma(Vector(p, 0.0), om));
  Ptr<ChisqModel> siginv(new ChisqModel(1.0, 0.5));
  Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(p, 0.5));
  Ptr<StateSpacePosteriorSampler> sampler(new StateSpacePosteriorSampler(&model, slab, siginv, spike));
  model.set_method(sampler);
  for (int i = 0; i < 150; ++i) model.sample_posterior();
  Matrix st = model.structural_state(5);
  EXPECT(st.nrow() == 2 + S - 1 && st.ncol() == T);
  // trend level + current seasonal effect track y - X coef
  double err = 0;
  for (int t = 0; t < T; ++t) {
    double target = y[t];
    for (int j = 0; j < p; ++j) target -= X(t, j) * coef[j];
    err += std::fabs(st(0, t) + st(2, t) - target);
  }
  EXPECT(err / T < 0.5);
  Vector v = model.state_variances(5);
  EXPECT(v[0] > 0 && v[1] > 0 && v[2] > 0 && v[1] < v[0] + 1.0);
}

// ... the general form: add_state in any order -- a weekly pattern of daily data whose
// "season" lasts a week, beside a day-of-week pattern, and no trend block at all but a local
// level added LAST (bsts: AddSeasonal(nseasons = 4, season.duration = 7) + AddSeasonal(7) +
// AddLocalLevel); the draws come from the look-ahead (the sampler's default)
static void AnyStateList() {
  const int T = 560, p = 3, chains = 6, D = 7, W = 4;
  std::mt19937_64 gen(23);
  std::normal_distribution<double> N(0, 1);
  Matrix X(T, p);
  Vector y(T), coef = {3.0, 0.0, -2.0};
  const double dow[D] = {1.0, -0.4, 0.2, -0.9, 0.6, -0.3, -0.2}, wk[W] = {0.8, -0.5, -0.6, 0.3};
  double level = 0;
  for (int t = 0; t < T; ++t) {
    level += 0.03 * N(gen);
    double mu = level + dow[t % D] + wk[(t / D) % W];
    for (int j = 0; j < p; ++j) { X(t, j) = N(gen); mu += X(t, j) * coef[j]; }
    y[t] = mu + 0.15 * N(gen);
  }
  StateSpaceRegressionModel model(y, X, std::vector<bool>(), chains, 11);
  Ptr<SeasonalStateModel> weekly(new SeasonalStateModel(W, D));
  weekly->set_sigsq(0.01);
  weekly->set_initial_state_variance(4.0);
  weekly->set_prior(1.0, 0.05);
  model.add_state(weekly);
  Ptr<SeasonalStateModel> daily(new SeasonalStateModel(D));
  daily->set_sigsq(0.01);
  daily->set_initial_state_variance(4.0);
  daily->set_prior(1.0, 0.05);
  model.add_state(daily);
  Ptr<LocalLevelStateModel> lv(new LocalLevelStateModel(0.3));
  lv->set_initial_state_mean(y[0]);
  lv->set_initial_state_variance(4.0);
  lv->set_prior(1.0, 0.1);
  model.add_state(lv);
  EXPECT(model.number_of_state_models() == 3);
  EXPECT(model.state_dimension() == (W - 1) + (D - 1) + 1);
  SpdMatrix om(p, p, 0.0);
  for (int j = 0; j < p; ++j) om(j, j) = 0.01;
  Ptr<MvnGivenScalarSigma> slab(new MvnGivenScalarSigma(Vector(p, 0.0), om));
  Ptr<ChisqModel> siginv(new ChisqModel(1.0, 0.5));
  Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(p, 0.5));
  Ptr<StateSpacePosteriorSampler> sampler(new StateSpacePosteriorSampler(&model, slab, siginv, spike));
  model.set_method(sampler);
  for (int i = 0; i < 200; ++i) model.sample_posterior();
  Matrix st = model.structural_state(0);
  EXPECT(st.nrow() == 10 && st.ncol() == T);
  // weekly effect + day-of-week effect + level track y - X coef
  double err = 0;
  for (int t = 0; t < T; ++t) {
    double target = y[t];
    for (int j = 0; j < p; ++j) target -= X(t, j) * coef[j];
    err += std::fabs(st(0, t) + st(W - 1, t) + st(W - 1 + D - 1, t) - target);
  }
  EXPECT(err / T < 0.4);
  // the weekly component holds still inside a week
  int moved = 0;
  for (int t = 1; t < T; ++t) moved += (t % D != 0 && st(0, t) != st(0, t - 1)) ? 1 : 0;
  EXPECT(moved == 0);
  Vector v = model.state_variances(0);
  EXPECT(v.size() == 3 && v[0] > 0 && v[1] > 0 && v[2] > 0);
}

// ... with an autoregressive component (bsts AddAr) on top of a local level
static void AutoregressiveState() {
  const int T = 600, p = 3, chains = 8;
  std::mt19937_64 gen(17);
  std::normal_distribution<double> N(0, 1);
  Matrix X(T, p);
  Vector y(T), coef = {2.0, 0.0, -1.5};
  double level = 0, u1 = 0, u2 = 0;
  for (int t = 0; t < T; ++t) {
    level += 0.02 * N(gen);
    const double u = 1.1 * u1 - 0.4 * u2 + 0.5 * N(gen);
    u2 = u1; u1 = u;
    double mu = level + u;
    for (int j = 0; j < p; ++j) { X(t, j) = N(gen); mu += X(t, j) * coef[j]; }
    y[t] = mu + 0.1 * N(gen);
  }
  StateSpaceRegressionModel model(y, X, std::vector<bool>(), chains, 5);
  Ptr<LocalLevelStateModel> level_model(new LocalLevelStateModel(0.1));
  level_model->set_initial_state_mean(y[0]);
  level_model->set_initial_state_variance(4.0);
  level_model->set_prior(1.0, 0.05, 0.2);
  model.add_state(level_model);
  Ptr<ArStateModel> ar(new ArStateModel(2));
  ar->set_sigma(0.5);
  ar->set_initial_state_variance(2.0);
  ar->set_prior(1.0, 0.5);
  model.add_state(ar);
  EXPECT(model.state_dimension() == 3);
  SpdMatrix om(p, p, 0.0);
  for (int j = 0; j < p; ++j) om(j, j) = 0.01;
  Ptr<MvnGivenScalarSigma> slab(new MvnGivenScalarSigma(Vector(p, 0.0), om));
  Ptr<ChisqModel> siginv(new ChisqModel(1.0, 0.5));
  Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(p, 0.5));
  Ptr<StateSpacePosteriorSampler> sampler(new StateSpacePosteriorSampler(&model, slab, siginv, spike));
  model.set_method(sampler);
  for (int i = 0; i < 300; ++i) model.sample_posterior();
  double p1 = 0, p2 = 0, s2 = 0;
  for (int c = 0; c < chains; ++c) {
    const Vector phi = model.ar_phi(c);
    EXPECT(phi.size() == 2);
    p1 += phi[0] / chains; p2 += phi[1] / chains; s2 += model.ar_sigsq(c) / chains;
  }
  // the data's autoregression (1.1, -0.4; innovation variance 0.25) is found
  EXPECT(std::fabs(p1 - 1.1) < 0.25 && std::fabs(p2 + 0.4) < 0.25);
  EXPECT(s2 > 0.1 && s2 < 0.6);
  Matrix st = model.structural_state(2);
  EXPECT(st.nrow() == 3 && st.ncol() == T);
}

// ... a static intercept and a harmonic (trig) component -- round 6, SURVEY 8f-2's glob: the
// shape of Models/StateSpace/StateModels/tests/Trig_test.cc (a sinusoid of known period)
static void TrigAndStaticInterceptState() {
  const int T = 480, p = 3, chains = 8;
  std::mt19937_64 gen(29);
  std::normal_distribution<double> N(0, 1);
  Matrix X(T, p);
  Vector y(T), coef = {1.5, 0.0, -2.0};
  const double period = 24.0, icpt = 7.5;
  for (int t = 0; t < T; ++t) {
    const double w = 2 * 3.141592653589793 * t / period;
    double mu = icpt + 2.0 * std::cos(w) - 1.2 * std::sin(w) + 0.7 * std::cos(2 * w);
    for (int j = 0; j < p; ++j) { X(t, j) = N(gen); mu += X(t, j) * coef[j]; }
    y[t] = mu + 0.2 * N(gen);
  }
  StateSpaceRegressionModel model(y, X, std::vector<bool>(), chains, 13);
  Ptr<StaticInterceptStateModel> intercept(new StaticInterceptStateModel);
  intercept->set_initial_state_mean(y[0]);
  intercept->set_initial_state_variance(25.0);
  model.add_state(intercept);
  Ptr<TrigStateModel> trig(new TrigStateModel(period, Vector{1.0, 2.0}));
  trig->set_sigsq(0.01);
  trig->set_initial_state_variance(Vector(4, 9.0));
  trig->set_prior(1.0, 0.02, 0.5);
  model.add_state(trig);
  EXPECT(model.number_of_state_models() == 2);
  EXPECT(model.state_dimension() == 5);
  SpdMatrix om(p, p, 0.0);
  for (int j = 0; j < p; ++j) om(j, j) = 0.01;
  Ptr<MvnGivenScalarSigma> slab(new MvnGivenScalarSigma(Vector(p, 0.0), om));
  Ptr<ChisqModel> siginv(new ChisqModel(1.0, 0.5));
  Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(p, 0.5));
  Ptr<StateSpacePosteriorSampler> sampler(new StateSpacePosteriorSampler(&model, slab, siginv, spike));
  model.set_method(sampler);
  for (int i = 0; i < 300; ++i) model.sample_posterior();
  for (int c : {0, chains - 1}) {
    Matrix st = model.structural_state(c);
    EXPECT(st.nrow() == 5 && st.ncol() == T);
    // the intercept does not move in time (T = 1, no state error) and is the series' level
    int moved = 0;
    for (int t = 1; t < T; ++t) moved += std::fabs(st(0, t) - st(0, 0)) > 1e-9 * std::fabs(st(0, 0)) ? 1 : 0;
    EXPECT(moved == 0);
    EXPECT(std::fabs(st(0, 0) - icpt) < 0.3);
    // intercept + the two harmonics' first components track y - X coef
    double err = 0;
    for (int t = 0; t < T; ++t) {
      double target = y[t];
      for (int j = 0; j < p; ++j) target -= X(t, j) * coef[j];
      err += std::fabs(st(0, t) + st(1, t) + st(3, t) - target);
    }
    EXPECT(err / T < 0.3);
  }
  Vector v = model.state_variances(0);   // (the trig model's one variance; the intercept has none)
  EXPECT(v.size() == 1 && v[0] > 0 && v[0] <= 0.25);
}

// the logit / probit spike-and-slab samplers in the reference's shape:
// model.set_method(new BinomialLogitSpikeSlabSampler(&model, slab, spike))
template <class MODEL, class SAMPLER>
static void BinomialSpikeSlab(bool logit) {
  const int n = 3000, p = 12, chains = 16;
  std::mt19937_64 gen(5);
  std::normal_distribution<double> N(0, 1);
  std::uniform_real_distribution<double> U(0, 1);
  Matrix X(n, p);
  Vector y(n), nt(n, 1.0), coef(p, 0.0);
  coef[0] = 0.3; coef[1] = 1.4; coef[2] = -1.1;
  for (int i = 0; i < n; ++i) {
    double eta = 0;
    for (int j = 0; j < p; ++j) { X(i, j) = j == 0 ? 1.0 : N(gen); eta += X(i, j) * coef[j]; }
    const double pr = logit ? 1 / (1 + std::exp(-eta)) : 0.5 * std::erfc(-eta / std::sqrt(2.0));
    y[i] = U(gen) < pr ? 1.0 : 0.0;
  }
  MODEL model(X, y, nt, 5, chains, 9);
  SpdMatrix prec(p, p, 0.0);
  for (int j = 0; j < p; ++j) prec(j, j) = 0.5;
  Ptr<MvnModel> slab(new MvnModel(Vector(p, 0.0), prec));
  Ptr<VariableSelectionPrior> spike(new VariableSelectionPrior(p, 3.0 / p));
  Ptr<SAMPLER> sampler(new SAMPLER(&model, slab, spike));
  model.set_method(sampler);
  model.drop_all();
  model.add(0);
  Vector inclusion(p, 0.0), mean(p, 0.0);
  const int burn = 60, niter = 140;
  for (int i = 0; i < burn + niter; ++i) {
    model.sample_posterior();
    if (i >= burn)
      for (int j = 0; j < p; ++j) { inclusion[j] += model.inc()[j] / (double)niter; mean[j] += model.Beta()[j] / niter; }
  }
  EXPECT(inclusion[1] > 0.95 && inclusion[2] > 0.95);
  for (int j = 3; j < p; ++j) EXPECT(inclusion[j] < 0.5);
  EXPECT(std::fabs(mean[1] - coef[1]) < 0.25 && std::fabs(mean[2] - coef[2]) < 0.25);
}

// The reference's callers draw once per iteration and record the model's
// parameters; with a look-ahead the sampler runs many sweeps per launch and
// serves them one by one: the recorded sequence must be the same.
static void LookAhead() {
  const int n = 400, p = 12, chains = 4, niter = 45;
  Vector beta(p, 0.0);
  beta[0] = 1.0; beta[3] = -2.0; beta[7] = 1.5;
  Sim s = simulate(n, p, beta, 1.0, 5);
  std::vector<double> rec[2];
  for (int mode = 0; mode < 2; ++mode) {
    RegressionModel model(s.X, s.y, chains, 99);
    Ptr<BregVsSampler> sampler(new BregVsSampler(&model, 1.0, 0.5, 3.0, true));
    model.set_method(sampler);
    model.drop_all();
    model.add(0);
    if (mode == 1) sampler->set_lookahead(20);
    for (int i = 0; i < niter; ++i) {
      model.sample_posterior();
      for (int j = 0; j < p; ++j) rec[mode].push_back(model.inc()[j] ? model.Beta()[j] : 0.0);
      rec[mode].push_back(model.sigsq());
    }
  }
  EXPECT(rec[0].size() == rec[1].size());
  bool same = true;
  for (size_t i = 0; i < rec[0].size() && i < rec[1].size(); ++i) same = same && (rec[0][i] == rec[1][i]);
  EXPECT(same);
}

// A device list behind the same classes (Engine over ba_group_*): the list repeats
// device 0 so that a one-GPU box runs it.  Global chain c of the list is chain c of one
// engine with all the chains (the sufficient statistics come from row shards summed in
// another order, hence a tolerance instead of equality).
static void DeviceList() {
  const int n = 500, p = 14, per = 3, niter = 30;
  Vector beta(p, 0.0);
  beta[0] = 1.0; beta[2] = -1.5; beta[9] = 2.0;
  Sim s = simulate(n, p, beta, 1.0, 21);
  std::vector<uint8_t> g[2];
  Vector b[2], s2[2];
  std::vector<double> rec[2];
  for (int mode = 0; mode < 2; ++mode) {
    Ptr<RegressionModel> model;
    if (mode == 0) model.reset(new RegressionModel(s.X, s.y, 2 * per, 31));
    else model.reset(new RegressionModel(s.X, s.y, per, std::vector<int>{0, 0}, 31));
    Ptr<BregVsSampler> sampler(new BregVsSampler(model.get(), 1.0, 0.5, 3.0, true));
    sampler->set_lookahead(8);
    sampler->set_correlation_swap_threshold(0.7);
    model->set_method(sampler);
    model->drop_all();
    model->add(0);
    for (int i = 0; i < niter; ++i) {
      model->sample_posterior();
      rec[mode].push_back(model->sigsq());
    }
    model->chain_states(g[mode], b[mode], s2[mode]);
  }
  EXPECT(g[0].size() == (size_t)2 * per * p && g[0] == g[1]);
  bool close = b[0].size() == b[1].size() && s2[0].size() == s2[1].size();
  for (size_t i = 0; close && i < b[0].size(); ++i)
    close = std::fabs(b[0][i] - b[1][i]) <= 1e-8 * std::max(1.0, std::fabs(b[0][i]));
  for (size_t i = 0; close && i < s2[0].size(); ++i) close = std::fabs(s2[0][i] - s2[1][i]) <= 1e-8 * s2[0][i];
  for (size_t i = 0; close && i < rec[0].size(); ++i) close = std::fabs(rec[0][i] - rec[1][i]) <= 1e-8 * rec[0][i];
  EXPECT(close);
}

// The virtual surface (draw / logpri / set_seed through a base pointer), ctor #4
// (ZellnerPriorParameters) against ctor #3 with the same numbers, and a change of
// the model's parameters between draws reaching the chains.
static void VirtualSurfaceAndCtor4() {
  const int n = 300, p = 10, chains = 3;
  Vector beta(p, 0.0);
  beta[0] = 1.0; beta[2] = 2.0;
  Sim s = simulate(n, p, beta, 1.0, 9);
  Vector b(p, 0.0), pi(p, 0.3);
  pi[0] = 1.0;
  SpdMatrix om(p, p, 0.0);
  for (int j = 0; j < p; ++j) om(j, j) = 0.5;
  ZellnerPriorParameters zp;
  zp.prior_inclusion_probabilities = pi;
  zp.prior_beta_guess = b;
  zp.prior_beta_guess_weight = 1.0;
  zp.prior_beta_information = om;
  zp.prior_sigma_guess = 1.1;
  zp.prior_sigma_guess_weight = 2.0;
  std::vector<double> rec[2];
  for (int mode = 0; mode < 2; ++mode) {
    RegressionModel model(s.X, s.y, chains, 7);
    Ptr<BregVsSampler> bvs;
    if (mode == 0) bvs.reset(new BregVsSampler(&model, zp));
    else bvs.reset(new BregVsSampler(&model, b, om, 1.1, 2.0, pi));
    Ptr<PosteriorSampler> sampler = bvs;       // the callers only ever see this
    model.set_method(sampler);
    model.drop_all();
    model.add(0);
    for (int i = 0; i < 12; ++i) {
      model.sample_posterior();
      rec[mode].push_back(model.sigsq());
      rec[mode].push_back(model.sampler(0)->logpri());
      if (i == 5) {
        // the caller changes the model between draws; with model selection
        // switched off for the next draw the change must still be there after it
        model.drop_all();
        model.add(0);
        model.add(4);
        model.set_sigsq(2.0);
        bvs->suppress_model_selection();
        model.sample_posterior();
        EXPECT(model.inc()[0] && model.inc()[4] && model.inc().nvars() == 2);
        EXPECT(model.Beta()[4] != 0.0);
        rec[mode].push_back(model.Beta()[4]);
        bvs->allow_model_selection();
      }
      if (i == 8) model.sampler(0)->set_seed(1234);
    }
    EXPECT(std::isfinite(rec[mode].back()));
  }
  bool same = rec[0].size() == rec[1].size();
  for (size_t i = 0; same && i < rec[0].size(); ++i) same = rec[0][i] == rec[1][i];
  EXPECT(same);
}

int main() {
  try {
    Small();
    LookAhead();
    VirtualSurfaceAndCtor4();
    DeviceList();
    TestMaxSizeControl();
    Large();
    PerfectCollinearity();
    ErrorConventions();
    StateSpace();
    StructuralTimeSeries();
    AutoregressiveState();
    TrigAndStaticInterceptState();
    AnyStateList();
    BinomialSpikeSlab<BinomialLogitModel, BinomialLogitSpikeSlabSampler>(true);
    BinomialSpikeSlab<BinomialProbitModel, BinomialProbitSpikeSlabSampler>(false);
  } catch (std::exception &e) {
    std::printf("EXCEPTION: %s\n", e.what());
    return 2;
  }
  if (failures) {
    std::printf("%d check(s) failed\n", failures);
    return 1;
  }
  std::printf("ALL OK\n");
  return 0;
}
