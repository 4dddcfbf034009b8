// See DeviceBinomialLogitSpikeSlabSampler.hpp.
#include "DeviceBinomialLogitSpikeSlabSampler.hpp"

#include <vector>

#include "LinAlg/SpdMatrix.hpp"
#include "cpputil/report_error.hpp"
#include "distributions/rng.hpp"

namespace BOOM {

  DeviceBinomialLogitSpikeSlabSampler::DeviceBinomialLogitSpikeSlabSampler(
      BinomialLogitModel *model, const Ptr<MvnBase> &slab,
      const Ptr<VariableSelectionPrior> &spike, int clt_threshold, int chains, int device,
      RNG &seeding_rng)
      : PosteriorSampler(seeding_rng),
        model_(model),
        slab_(slab),
        engine_(nullptr),
        chains_(chains) {
    const int p = model->xdim();
    if (static_cast<int>(slab->dim()) != p) report_error("Slab does not match model dimension.");
    if (static_cast<int>(spike->potential_nvars()) != p) report_error("Spike does not match model dimension.");
    device_seed_ = seed_rng(seeding_rng);
    ba_config cfg{device, chains, 0, static_cast<uint64_t>(device_seed_), 0, 0};
    check(ba_engine_create(&cfg, &engine_));
    engines_.assign(1, engine_);
    try {   // (report_error throws: the engine must not leak)
      configure(spike, clt_threshold);
    } catch (...) {
      ba_engine_destroy(engine_);
      engine_ = nullptr;
      throw;
    }
  }

  DeviceBinomialLogitSpikeSlabSampler::DeviceBinomialLogitSpikeSlabSampler(
      BinomialLogitModel *model, const Ptr<MvnBase> &slab,
      const Ptr<VariableSelectionPrior> &spike, int clt_threshold, int chains_per_device,
      const std::vector<int> &devices, RNG &seeding_rng)
      : PosteriorSampler(seeding_rng),
        model_(model),
        slab_(slab),
        engine_(nullptr),
        chains_(chains_per_device * static_cast<int>(devices.size())) {
    const int p = model->xdim();
    if (devices.empty()) report_error("The device list is empty.");
    if (static_cast<int>(slab->dim()) != p) report_error("Slab does not match model dimension.");
    if (static_cast<int>(spike->potential_nvars()) != p) report_error("Spike does not match model dimension.");
    device_seed_ = seed_rng(seeding_rng);
    std::vector<int32_t> dv(devices.begin(), devices.end());
    if (ba_group_create(dv.data(), static_cast<int32_t>(dv.size()), chains_per_device,
                        static_cast<uint64_t>(device_seed_), &group_) != BA_OK)
      report_error(ba_group_last_error());
    for (int i = 0; i < ba_group_size(group_); ++i) engines_.push_back(ba_group_engine(group_, i));
    engine_ = engines_[0];
    try {
      configure(spike, clt_threshold);
    } catch (...) {
      ba_group_destroy(group_);
      group_ = nullptr;
      engine_ = nullptr;
      throw;
    }
  }

  void DeviceBinomialLogitSpikeSlabSampler::configure(const Ptr<VariableSelectionPrior> &spike,
                                                      int clt_threshold) {
    const int p = model_->xdim();
    // model->dat(): one BinomialRegressionData per observation -> column-major X, y, n
    const std::vector<Ptr<BinomialRegressionData>> &data(model_->dat());
    const size_t n = data.size();
    std::vector<double> X(n * p), y(n), nt(n);
    for (size_t i = 0; i < n; ++i) {
      const Vector &x(data[i]->x());
      for (int j = 0; j < p; ++j) X[static_cast<size_t>(j) * n + i] = x[j];
      y[i] = data[i]->y();
      nt[i] = data[i]->n();
    }
    const Vector mu = slab_->mu();
    const SpdMatrix siginv = slab_->siginv();
    const Vector pi = spike->prior_inclusion_probabilities();
    for (ba_engine *e : engines_) {
      check(ba_logit_set_data(e, static_cast<int64_t>(n), p, X.data(), y.data(), nt.data(), clt_threshold));
      check(ba_sss_set_slab(e, mu.data(), siginv.data(), 0, -1));
      check(ba_set_spike(e, pi.data(), spike->max_model_size()));
    }
    push_state();
  }

  DeviceBinomialLogitSpikeSlabSampler::~DeviceBinomialLogitSpikeSlabSampler() {
    if (group_) ba_group_destroy(group_);
    else if (engine_) ba_engine_destroy(engine_);
  }

  ba_engine *DeviceBinomialLogitSpikeSlabSampler::locate(int chain, int64_t *local) const {
    if (!group_) {
      *local = chain;
      return engine_;
    }
    int32_t ei = 0;
    if (ba_group_locate(group_, chain, &ei, local) != BA_OK) report_error(ba_group_last_error());
    return engines_[ei];
  }

  void DeviceBinomialLogitSpikeSlabSampler::check(int rc) const {
    if (rc != BA_OK) report_error(ba_last_error());
  }

  void DeviceBinomialLogitSpikeSlabSampler::draw() {
    for (ba_engine *e : engines_) check(ba_logit_sweep(e, 1));   // (every device's round is out before any is read)
    pull_chain0();
  }

  double DeviceBinomialLogitSpikeSlabSampler::logpri() const {
    report_error("logpri() is not implemented for DeviceBinomialLogitSpikeSlabSampler");
    return negative_infinity();
  }

  void DeviceBinomialLogitSpikeSlabSampler::limit_model_selection(int max_flips) {
    const Vector mu = slab_->mu();
    const SpdMatrix siginv = slab_->siginv();
    for (ba_engine *e : engines_) check(ba_sss_set_slab(e, mu.data(), siginv.data(), 0, max_flips > 0 ? max_flips : -1));
  }

  void DeviceBinomialLogitSpikeSlabSampler::push_state() {
    const Selector &inc(model_->coef().inc());
    const int p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    for (int j = 0; j < p; ++j) gamma[j] = inc[j] ? 1 : 0;
    const Vector beta = model_->Beta();
    for (ba_engine *e : engines_) check(ba_set_state(e, -1, gamma.data(), beta.data(), 1.0));
  }

  void DeviceBinomialLogitSpikeSlabSampler::chain_state(int chain, Selector &inc,
                                                        Vector &beta) const {
    const int p = model_->xdim();
    std::vector<uint8_t> gamma(p, 0);
    beta.resize(p);
    int64_t lc = 0;
    ba_engine *eng = locate(chain, &lc);
    check(ba_get_state(eng, lc, gamma.data(), beta.data(), nullptr));
    inc = Selector(p, false);
    for (int j = 0; j < p; ++j)
      if (gamma[j]) inc.add(j);
  }

  void DeviceBinomialLogitSpikeSlabSampler::pull_chain0() {
    Selector inc(model_->xdim(), false);
    Vector beta;
    chain_state(0, inc, beta);
    model_->coef().set_inc(inc);
    model_->coef().set_included_coefficients(inc.select(beta));
  }

}  // namespace BOOM
