// The reference-side binding for the logit sampler: a BOOM
// PosteriorSampler with BinomialLogitSpikeSlabSampler's constructor
// (BinomialLogitSpikeSlabSampler.hpp:29-33) plus a chain count, forwarding draw() to
// ba_logit_sweep through the C-ABI.  OUR code, written against the reference's public
// headers and compiled only where /root/reference exists (bindings/boom/Makefile;
// oracle/Makefile target `binding` links the test driver);
// bindings/boom/binding_driver.cpp runs it under the reference's own
// `model->sample_posterior()` loop.  It is what a BOOM maintainer would add under
// Models/Glm/PosteriorSamplers/.
#ifndef BOOM_AMD_DEVICE_BINOMIAL_LOGIT_SPIKE_SLAB_SAMPLER_HPP_
#define BOOM_AMD_DEVICE_BINOMIAL_LOGIT_SPIKE_SLAB_SAMPLER_HPP_

#include "LinAlg/Selector.hpp"
#include "LinAlg/Vector.hpp"
#include "Models/Glm/BinomialLogitModel.hpp"
#include "Models/Glm/VariableSelectionPrior.hpp"
#include "Models/MvnBase.hpp"
#include "Models/PosteriorSamplers/PosteriorSampler.hpp"

#include <vector>

extern "C" {
#include "boom_amd.h"
}

namespace BOOM {

  // Chain 0 backs the model's coefficients (coef().inc(), Beta()) between draws; the
  // other chains are read with chain_state().  The data (model->dat()) go to the device
  // once, in the constructor: add the data before creating the sampler.
  class DeviceBinomialLogitSpikeSlabSampler : public PosteriorSampler {
   public:
    DeviceBinomialLogitSpikeSlabSampler(BinomialLogitModel *model, const Ptr<MvnBase> &slab,
                                        const Ptr<VariableSelectionPrior> &spike,
                                        int clt_threshold, int chains, int device = 0,
                                        RNG &seeding_rng = GlobalRng::rng);
    // ... over several devices behind one handle (ba_group_*): chains_per_device chains on
    // every entry of `devices`, global chain ids device-major, the data replicated
    DeviceBinomialLogitSpikeSlabSampler(BinomialLogitModel *model, const Ptr<MvnBase> &slab,
                                        const Ptr<VariableSelectionPrior> &spike,
                                        int clt_threshold, int chains_per_device,
                                        const std::vector<int> &devices,
                                        RNG &seeding_rng = GlobalRng::rng);
    ~DeviceBinomialLogitSpikeSlabSampler() override;

    void draw() override;            // BinomialLogitSpikeSlabSampler::draw, .cpp:50-54
    double logpri() const override;  // not on the device: reported
    void limit_model_selection(int max_flips);   // BinomialLogitSpikeSlabSampler.hpp:56

    unsigned long device_seed() const { return device_seed_; }
    int number_of_chains() const { return chains_; }
    void chain_state(int chain, Selector &inc, Vector &beta) const;

   private:
    void check(int rc) const;
    void push_state();
    void pull_chain0();
    void configure(const Ptr<VariableSelectionPrior> &spike, int clt_threshold);
    ba_engine *locate(int chain, int64_t *local) const;
    BinomialLogitModel *model_;
    Ptr<MvnBase> slab_;
    ba_engine *engine_;               // chain 0's engine
    ba_group *group_ = nullptr;       // the device list's handle (nullptr: one engine)
    std::vector<ba_engine *> engines_;
    int chains_;                      // all chains
    unsigned long device_seed_;
  };

}  // namespace BOOM
#endif  // BOOM_AMD_DEVICE_BINOMIAL_LOGIT_SPIKE_SLAB_SAMPLER_HPP_
