// See DevicePoissonRegressionSpikeSlabSampler.hpp.
#include "DevicePoissonRegressionSpikeSlabSampler.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

#include "LinAlg/SpdMatrix.hpp"
#include "Models/Glm/PosteriorSamplers/NormalMixtureApproximation.hpp"
#include "Models/Glm/PosteriorSamplers/poisson_mixture_approximation_table.hpp"
#include "cpputil/report_error.hpp"
#include "distributions/rng.hpp"

namespace BOOM {

  DevicePoissonRegressionSpikeSlabSampler::DevicePoissonRegressionSpikeSlabSampler(
      PoissonRegressionModel *model, const Ptr<MvnBase> &slab,
      const Ptr<VariableSelectionPrior> &spike, int chains, int device, RNG &seeding_rng)
      : PosteriorSampler(seeding_rng), model_(model), slab_(slab), engine_(nullptr), chains_(chains) {
    const int p = model->xdim();
    if (static_cast<int>(slab->dim()) != p) report_error("Slab does not match model dimension.");
    if (static_cast<int>(spike->potential_nvars()) != p) report_error("Spike does not match model dimension.");
    device_seed_ = seed_rng(seeding_rng);
    ba_config cfg{device, chains, 0, static_cast<uint64_t>(device_seed_), 0, 0};
    check(ba_engine_create(&cfg, &engine_));
    try {   // (report_error throws and a throwing constructor runs no destructor: the engine must not leak)
    // model->dat(): one PoissonRegressionData per observation -> column-major X, y, exposure
    const std::vector<Ptr<PoissonRegressionData>> &data(model->dat());
    const size_t n = data.size();
    std::vector<double> X(n * p), y(n), exposure(n);
    for (size_t i = 0; i < n; ++i) {
      const Vector &x(data[i]->x());
      for (int j = 0; j < p; ++j) X[static_cast<size_t>(j) * n + i] = x[j];
      y[i] = data[i]->y();
      exposure[i] = data[i]->exposure();
    }
    check(ba_poisson_set_data(engine_, static_cast<int64_t>(n), p, X.data(), y.data(), exposure.data()));
    // The mixtures: BOOM's table, asked the way PoissonDataImputer::impute asks in a pass
    // over the data (count 1 for the event past the interval, then the observation's own
    // count): approximate() refits and grows the table on demand, so the order matters.
    NormalMixtureApproximationTable table = create_poisson_mixture_approximation_table();
    const int64_t largest = table.largest_index();
    std::vector<int64_t> counts;
    for (size_t i = 0; i < n; ++i) {
      table.approximate(1);
      const int64_t c = llround(y[i]);
      if (c > 0 && c < largest) table.approximate(static_cast<int>(c));
      if (c > 0 && c < largest) counts.push_back(c);
    }
    counts.push_back(1);
    std::sort(counts.begin(), counts.end());
    counts.erase(std::unique(counts.begin(), counts.end()), counts.end());
    std::vector<int32_t> ncomp;
    std::vector<double> mu, sigma, weight;
    for (int64_t c : counts) {
      const NormalMixtureApproximation &a(table.approximate(static_cast<int>(c)));
      ncomp.push_back(a.dim());
      for (int k = 0; k < a.dim(); ++k) {
        mu.push_back(a.mu()[k]);
        sigma.push_back(a.sigma()[k]);
        weight.push_back(a.weights()[k]);
      }
    }
    check(ba_poisson_set_mixtures(engine_, static_cast<int32_t>(counts.size()), counts.data(), ncomp.data(),
                                  mu.data(), sigma.data(), weight.data(), largest));
    const Vector smu = slab->mu();
    const SpdMatrix siginv = slab->siginv();
    check(ba_sss_set_slab(engine_, smu.data(), siginv.data(), 0, -1));
    const Vector pi = spike->prior_inclusion_probabilities();
    check(ba_set_spike(engine_, pi.data(), spike->max_model_size()));
    push_state();
    } catch (...) {
      ba_engine_destroy(engine_);
      engine_ = nullptr;
      throw;
    }
  }

  DevicePoissonRegressionSpikeSlabSampler::~DevicePoissonRegressionSpikeSlabSampler() {
    ba_engine_destroy(engine_);
  }

  void DevicePoissonRegressionSpikeSlabSampler::check(int rc) const {
    if (rc != BA_OK) report_error(ba_last_error());
  }

  void DevicePoissonRegressionSpikeSlabSampler::draw() {
    check(ba_poisson_sweep(engine_, 1));
    pull_chain0();
  }

  double DevicePoissonRegressionSpikeSlabSampler::logpri() const {
    report_error("logpri() is not implemented for DevicePoissonRegressionSpikeSlabSampler");
    return negative_infinity();
  }

  void DevicePoissonRegressionSpikeSlabSampler::limit_model_selection(int max_flips) {
    const Vector mu = slab_->mu();
    const SpdMatrix siginv = slab_->siginv();
    check(ba_sss_set_slab(engine_, mu.data(), siginv.data(), 0, max_flips > 0 ? max_flips : -1));
  }

  void DevicePoissonRegressionSpikeSlabSampler::push_state() {
    const Selector &inc(model_->coef().inc());
    const int p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    for (int j = 0; j < p; ++j) gamma[j] = inc[j] ? 1 : 0;
    const Vector beta = model_->Beta();
    check(ba_set_state(engine_, -1, gamma.data(), beta.data(), 1.0));
  }

  void DevicePoissonRegressionSpikeSlabSampler::chain_state(int chain, Selector &inc, Vector &beta) const {
    const int p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    beta.resize(p);
    check(ba_get_state(engine_, chain, gamma.data(), beta.data(), nullptr));
    inc = Selector(p, false);
    for (int j = 0; j < p; ++j)
      if (gamma[j]) inc.add(j);
  }

  void DevicePoissonRegressionSpikeSlabSampler::pull_chain0() {
    Selector inc(model_->xdim(), false);
    Vector beta;
    chain_state(0, inc, beta);
    model_->coef().set_inc(inc);
    model_->coef().set_included_coefficients(inc.select(beta));
  }

}  // namespace BOOM
