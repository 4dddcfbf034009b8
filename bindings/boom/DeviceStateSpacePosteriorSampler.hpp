// The reference-side binding for the bsts half of the path: a BOOM PosteriorSampler
// that stands where StateSpacePosteriorSampler stands
// (Models/StateSpace/PosteriorSamplers/StateSpacePosteriorSampler.hpp:27-33) for a
// StateSpaceRegressionModel with WHATEVER list of state models add_state gave it --
// LocalLevelStateModel, LocalLinearTrendStateModel, SeasonalStateModel (any season
// duration), ArStateModel, in any order and number (state dimension <= 64) -- and
// forwards draw() to the engine through the C-ABI (include/boom_amd.h).
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
#include "Models/GaussianModelBase.hpp"
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
    // For the SLOPE entry of a SemilocalLinearTrendStateModel only (null / ignored elsewhere): what
    // its NonzeroMeanAr1Sampler takes besides the variance prior (NonzeroMeanAr1Sampler.hpp: the
    // long-run mean's and the AR(1) coefficient's Gaussian priors, force_stationary(),
    // force_ar1_positive()), as bsts builds it (Interfaces/R/bsts/src/create_state_model.cpp:601-672).
    Ptr<GaussianModelBase> slope_mean_prior;
    Ptr<GaussianModelBase> slope_ar1_prior;
    bool force_stationary = true;
    bool force_ar1_positive = false;
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
  //       one entry per variance parameter, in the order the state models were added:
  //       (level) for a LocalLevelStateModel, (level, slope) for a
  //       LocalLinearTrendStateModel, (seasonal) for a SeasonalStateModel, the
  //       ArPosteriorSampler's prior for an ArStateModel, the error distribution's for a
  //       TrigStateModel, (level, slope -- with the slope's other priors) for a
  //       SemilocalLinearTrendStateModel; none for a StaticInterceptStateModel.
  //   seasonal_time_of_first_observation
  //       SeasonalStateModel keeps what set_time_of_first_observation was given to itself:
  //       one entry per SeasonalStateModel, in order (empty: all 0).
  // The data, the state models' initial-state distributions and the parameters' current
  // values are read from the model in the constructor: add the data and the state
  // models first.  Chain 0 backs the model's own objects: after every draw
  // regression_model()'s coefficients / inclusion indicators / sigsq, the state models'
  // variances (and coefficients) and model->state() hold chain 0's draw.
  // draw() is served from the engine's look-ahead (ba_ss_set_lookahead /
  // ba_ss_draw_next: rounds enqueued `lookahead` at a time, every round's draw recorded
  // on the device), which no caller can observe: the loop
  //     for (i in niter) { model->sample_posterior(); record }
  // (Interfaces/R/bsts/src/bsts.cc:82-119) runs at the device's rate.
  class DeviceStateSpacePosteriorSampler : public PosteriorSampler {
   public:
    DeviceStateSpacePosteriorSampler(
        StateSpaceRegressionModel *model,
        const Ptr<MvnGivenScalarSigmaBase> &slab,
        const Ptr<GammaModelBase> &residual_precision_prior,
        const Ptr<VariableSelectionPrior> &spike,
        double sigma_upper_limit,
        const std::vector<DeviceStateVariancePrior> &state_variance_priors,
        int chains, int device = 0, RNG &seeding_rng = GlobalRng::rng,
        const std::vector<int> &seasonal_time_of_first_observation = std::vector<int>(),
        int lookahead = 64);
    // ... over several devices behind one handle (ba_group_*): chains_per_device chains on
    // every entry of `devices`, global chain ids device-major; the data are replicated and
    // the chains never communicate, every device's rounds are enqueued before
    // any is waited for.  Chain 0 (device 0) backs the model.
    DeviceStateSpacePosteriorSampler(
        StateSpaceRegressionModel *model,
        const Ptr<MvnGivenScalarSigmaBase> &slab,
        const Ptr<GammaModelBase> &residual_precision_prior,
        const Ptr<VariableSelectionPrior> &spike,
        double sigma_upper_limit,
        const std::vector<DeviceStateVariancePrior> &state_variance_priors,
        int chains_per_device, const std::vector<int> &devices, RNG &seeding_rng = GlobalRng::rng,
        const std::vector<int> &seasonal_time_of_first_observation = std::vector<int>(),
        int lookahead = 64);
    ~DeviceStateSpacePosteriorSampler() override;

    void draw() override;            // StateSpacePosteriorSampler::draw, .cpp:42-64
    double logpri() const override;  // StateSpacePosteriorSampler::logpri, .cpp:66-74 (chain 0)

    unsigned long device_seed() const { return device_seed_; }
    void set_device_seed(unsigned long seed);

    int number_of_chains() const { return chains_; }
    int state_dimension() const { return state_dim_; }
    // one chain's autoregression coefficients and error variance (the model's first --
    // or `which`-th -- ArStateModel)
    void chain_ar(int chain, Vector &phi, double &sigsq, int which = 0) const;
    // the other chains: regression parameters, the state variances (one entry per
    // variance parameter, in state-model order -- state_variance_priors' order) and the
    // state draw (state_dimension x time_dimension)
    void chain_state(int chain, Selector &inc, Vector &beta, double &sigsq,
                     Vector &state_variances, Matrix &state) const;
    // whose state paths the look-ahead keeps besides chain 0's (reading another chain's
    // state is correct all the same, and slow)
    void record_state_of_chains(const std::vector<int> &chains);
    // StateSpaceRegressionModel::simulate_forecast(rng, newX, final_state)
    // (StateSpaceRegressionModel.cpp:214-219) for EVERY chain's current draw: one row
    // per chain, nrow(newX) columns
    Matrix simulate_forecast(const Matrix &newX);

   private:
    void check(int rc) const;
    void pull_chain0();
    void classify(const Ptr<MvnGivenScalarSigmaBase> &slab, const Ptr<VariableSelectionPrior> &spike);
    void configure(const Ptr<MvnGivenScalarSigmaBase> &slab, const Ptr<GammaModelBase> &residual_precision_prior,
                   const Ptr<VariableSelectionPrior> &spike, double sigma_upper_limit,
                   const std::vector<int> &seasonal_time_of_first_observation, int lookahead);
    // the engine that holds a (global) chain, and the chain's index there
    ba_engine *locate(int chain, int64_t *local) const;
    StateSpaceRegressionModel *model_;
    ba_engine *engine_;               // chain 0's engine (= engines_[0])
    ba_group *group_ = nullptr;       // the device list's handle (nullptr: one engine)
    std::vector<ba_engine *> engines_;
    int chains_;                      // all chains
    unsigned long device_seed_;
    // the state models as the engine knows them
    struct Block {
      int kind;    // 1 local level, 2 local linear trend, 3 seasonal, 4 autoregression, 5 static intercept, 6 trig, 7 semilocal linear trend
      int var0;    // index of its first variance parameter (into state_variance_priors)
      int nvar, dim, lags;
    };
    std::vector<Block> blocks_;
    int state_dim_;
    bool structural_;  // anything but a lone local level: the engine runs a state-model list
    std::vector<DeviceStateVariancePrior> variance_priors_;
    // The regression's three prior objects are the caller's (as BregVsSampler's ctor #5
    // takes them, BregVsSampler.hpp:98-106): the sampler observes their parameters
    // (Data::add_observer, DataTypes.hpp:76) and a draw() that finds them changed uploads them
    // before it launches.  (The state models' variance priors are read at construction.)
    Ptr<MvnGivenScalarSigmaBase> slab_;
    Ptr<GammaModelBase> residual_precision_prior_;
    Ptr<VariableSelectionPrior> spike_;
    double sigma_upper_limit_ = 0;
    bool priors_stale_ = false;
    void upload_regression_priors();
    void observe();
    void unobserve();
   public:
    // the device copies of the regression's priors are stale (a change no parameter signals:
    // the slab's unscaled precision): the next draw() reads the objects again
    void refresh_priors() { priors_stale_ = true; }
  };

}  // namespace BOOM
#endif  // BOOM_AMD_DEVICE_STATE_SPACE_POSTERIOR_SAMPLER_HPP_
