// The reference-side binding for the bsts half of the path: a BOOM PosteriorSampler
// that stands where StateSpacePosteriorSampler stands
// (Models/StateSpace/PosteriorSamplers/StateSpacePosteriorSampler.hpp:27-33) for a
// StateSpaceRegressionModel whose state is a LocalLevelStateModel -- or a
// LocalLinearTrendStateModel / LocalLevelStateModel followed by a SeasonalStateModel --
// and forwards draw() to ba_ss_sweep through the C-ABI (include/boom_amd.h).
//
// OUR code, written against the reference's public headers.  It is compiled where the
// reference tree exists (bindings/boom/Makefile; oracle/Makefile target `binding` links
// it with the compiled reference and the product library), and
// bindings/boom/binding_driver.cpp runs it under the reference's own
// `model->sample_posterior()` loop on the GPU box (tests/test_reference_binding_gpu.py).
// It is what a BOOM maintainer would add under Models/StateSpace/PosteriorSamplers/.
#ifndef BOOM_AMD_DEVICE_STATE_SPACE_POSTERIOR_SAMPLER_HPP_
#define BOOM_AMD_DEVICE_STATE_SPACE_POSTERIOR_SAMPLER_HPP_

#include <vector>

#include "LinAlg/Matrix.hpp"
#include "LinAlg/Selector.hpp"
#include "LinAlg/Vector.hpp"
#include "Models/GammaModel.hpp"
#include "Models/Glm/VariableSelectionPrior.hpp"
#include "Models/MvnGivenScalarSigma.hpp"
#include "Models/PosteriorSamplers/PosteriorSampler.hpp"
#include "Models/StateSpace/StateModels/ArStateModel.hpp"
#include "Models/StateSpace/StateSpaceRegressionModel.hpp"

extern "C" {
#include "boom_amd.h"
}

namespace BOOM {

  // The prior of one state-model variance parameter, as the reference's samplers take it
  // (ZeroMeanGaussianConjSampler(model, Ptr<GammaModelBase>) +
  // set_sigma_upper_limit, ZeroMeanGaussianConjSampler.cpp:37-60;
  // ZeroMeanMvnIndependenceSampler likewise).
  struct DeviceStateVariancePrior {
    Ptr<GammaModelBase> precision_prior;
    double sigma_upper_limit;
  };

  // Many-chain drop-in for StateSpacePosteriorSampler on a StateSpaceRegressionModel.
  //
  // In the reference the observation model and every state model carry their own
  // samplers (BregVsSampler, ZeroMeanGaussianConjSampler, ...), and
  // StateSpacePosteriorSampler::draw() (StateSpacePosteriorSampler.cpp:42-64) calls them
  // in turn and then imputes the state.  Here the whole sweep is one call into the
  // engine, so the priors those samplers would hold are constructor arguments:
  //   slab, residual_precision_prior, spike, sigma_upper_limit
  //       BregVsSampler's ctor #5 on model->regression_model() + set_sigma_upper_limit;
  //   state_variance_priors
  //       one entry per variance parameter in state-model order: (level) for a
  //       LocalLevelStateModel, (level, slope) for a LocalLinearTrendStateModel, then
  //       (seasonal) if a SeasonalStateModel follows, then the ArPosteriorSampler's
  //       prior if an ArStateModel comes last.
  // The data, the state models' initial-state distributions and the parameters' current
  // values are read from the model in the constructor: add the data and the state
  // models first.  Chain 0 backs the model's own objects: after every draw
  // regression_model()'s coefficients / inclusion indicators / sigsq, the state models'
  // variances and model->state() hold chain 0's draw.
  class DeviceStateSpacePosteriorSampler : public PosteriorSampler {
   public:
    DeviceStateSpacePosteriorSampler(
        StateSpaceRegressionModel *model,
        const Ptr<MvnGivenScalarSigmaBase> &slab,
        const Ptr<GammaModelBase> &residual_precision_prior,
        const Ptr<VariableSelectionPrior> &spike,
        double sigma_upper_limit,
        const std::vector<DeviceStateVariancePrior> &state_variance_priors,
        int chains, int device = 0, RNG &seeding_rng = GlobalRng::rng);
    ~DeviceStateSpacePosteriorSampler() override;

    void draw() override;            // StateSpacePosteriorSampler::draw, .cpp:42-64
    double logpri() const override;  // StateSpacePosteriorSampler::logpri, .cpp:66-74 (chain 0)

    unsigned long device_seed() const { return device_seed_; }
    void set_device_seed(unsigned long seed);

    int number_of_chains() const { return chains_; }
    int state_dimension() const { return state_dim_; }
    // one chain's autoregression coefficients and error variance (an ArStateModel last)
    void chain_ar(int chain, Vector &phi, double &sigsq) const;
    // the other chains: regression parameters, state variances (level, slope,
    // seasonal; unused entries 0) and the state draw (state_dimension x time_dimension)
    void chain_state(int chain, Selector &inc, Vector &beta, double &sigsq,
                     Vector &state_variances, Matrix &state) const;
    // StateSpaceRegressionModel::simulate_forecast(rng, newX, final_state)
    // (StateSpaceRegressionModel.cpp:214-219) for EVERY chain's current draw: one row
    // per chain, nrow(newX) columns
    Matrix simulate_forecast(const Matrix &newX);

   private:
    void check(int rc) const;
    void pull_chain0();
    StateSpaceRegressionModel *model_;
    ba_engine *engine_;
    int chains_;
    unsigned long device_seed_;
    int trend_;      // 1 local level, 2 local linear trend
    int nseasons_;   // 0: no seasonal state model
    int state_dim_;
    int ar_index_ = -1;   // position of the ArStateModel among the state models (-1: none)
    int ar_lags_ = 0;
    bool structural_;  // the engine runs ba_ss_set_structural (anything but a lone local level)
    std::vector<DeviceStateVariancePrior> variance_priors_;
  };

}  // namespace BOOM
#endif  // BOOM_AMD_DEVICE_STATE_SPACE_POSTERIOR_SAMPLER_HPP_
