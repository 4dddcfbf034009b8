// The reference-side binding for the Poisson sampler: a BOOM
// PosteriorSampler with PoissonRegressionSpikeSlabSampler's constructor
// (PoissonRegressionSpikeSlabSampler.hpp:41-44) plus a chain count, forwarding draw() to
// ba_poisson_sweep through the C-ABI.  OUR code, written against the reference's public
// headers (bindings/boom/Makefile; oracle/Makefile target `binding` links the test
// driver).  It also shows where the engine's normal mixtures of NegLogGamma(count) come
// from on the BOOM side: the reference's own table
// (create_poisson_mixture_approximation_table + approximate), asked in the order the
// reference's imputer would ask.
#ifndef BOOM_AMD_DEVICE_POISSON_REGRESSION_SPIKE_SLAB_SAMPLER_HPP_
#define BOOM_AMD_DEVICE_POISSON_REGRESSION_SPIKE_SLAB_SAMPLER_HPP_

#include "LinAlg/Selector.hpp"
#include "LinAlg/Vector.hpp"
#include "Models/Glm/PoissonRegressionModel.hpp"
#include "Models/Glm/VariableSelectionPrior.hpp"
#include "Models/MvnBase.hpp"
#include "Models/PosteriorSamplers/PosteriorSampler.hpp"

extern "C" {
#include "boom_amd.h"
}

namespace BOOM {

  // Chain 0 backs the model's coefficients between draws; the other chains are read with
  // chain_state().  The data (model->dat()) go to the device once, in the constructor.
  class DevicePoissonRegressionSpikeSlabSampler : public PosteriorSampler {
   public:
    DevicePoissonRegressionSpikeSlabSampler(PoissonRegressionModel *model, const Ptr<MvnBase> &slab,
                                            const Ptr<VariableSelectionPrior> &spike, int chains,
                                            int device = 0, RNG &seeding_rng = GlobalRng::rng);
    ~DevicePoissonRegressionSpikeSlabSampler() override;

    void draw() override;            // PoissonRegressionSpikeSlabSampler::draw, .cpp:55-59
    double logpri() const override;  // not on the device: reported
    void limit_model_selection(int max_flips);

    unsigned long device_seed() const { return device_seed_; }
    int number_of_chains() const { return chains_; }
    void chain_state(int chain, Selector &inc, Vector &beta) const;

   private:
    void check(int rc) const;
    void push_state();
    void pull_chain0();
    PoissonRegressionModel *model_;
    Ptr<MvnBase> slab_;
    ba_engine *engine_;
    int chains_;
    unsigned long device_seed_;
  };

}  // namespace BOOM
#endif  // BOOM_AMD_DEVICE_POISSON_REGRESSION_SPIKE_SLAB_SAMPLER_HPP_
