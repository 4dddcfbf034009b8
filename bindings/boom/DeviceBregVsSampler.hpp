// The reference-side binding INTEGRATION.md sec. 1 describes, as a real file:
// a BOOM PosteriorSampler that forwards draw() to the boom_amd engine through
// the C-ABI (include/boom_amd.h).  This is OUR code, written against the
// reference's public headers; it is compiled only where /root/reference exists
// (bindings/boom/Makefile; oracle/Makefile target `binding` links the test
// driver) so that the boundary claim is checked by a compiler and -- through
// bindings/boom/binding_driver.cpp -- exercised on the GPU box by the
// reference's own `model->sample_posterior()` loop.  It is what a BOOM
// maintainer would add under Models/Glm/PosteriorSamplers/.
#ifndef BOOM_AMD_DEVICE_BREG_VS_SAMPLER_HPP_
#define BOOM_AMD_DEVICE_BREG_VS_SAMPLER_HPP_

#include <vector>
#include "LinAlg/Selector.hpp"
#include "LinAlg/Vector.hpp"
#include "Models/GammaModel.hpp"
#include "Models/Glm/PosteriorSamplers/BregVsSampler.hpp"   // ZellnerPriorParameters
#include "Models/Glm/RegressionModel.hpp"
#include "Models/Glm/VariableSelectionPrior.hpp"
#include "Models/MvnGivenScalarSigma.hpp"
#include "Models/PosteriorSamplers/PosteriorSampler.hpp"

extern "C" {
#include "boom_amd.h"
}

namespace BOOM {

  // Many-chain drop-in for BregVsSampler (same constructor arguments as its
  // ctor #5, BregVsSampler.hpp:102-106, plus the chain count).  Chain 0 backs
  // the model's own parameters, so code that reads model->Beta() / sigsq() /
  // coef().inc() between draws keeps working; the other chains are read with
  // chain_state().
  class DeviceBregVsSampler : public PosteriorSampler {
   public:
    // BregVsSampler's five constructors (BregVsSampler.hpp:64-106), each with the chain
    // count, the device and the look-ahead behind the reference's own arguments.
    // #1 (BregVsSampler.cpp:48-85) and #2 (:87-142) assemble slab, spike and residual prior
    // from the model's sufficient statistics -- on the engine (ba_set_priors_ctor1 / _ctor2);
    // the prior OBJECTS the sampler then holds are built from what the engine assembled.
    DeviceBregVsSampler(RegressionModel *model, double prior_nobs, double expected_rsq,
                        double expected_model_size, bool first_term_is_intercept,
                        int chains, int device = 0, int lookahead = 256,
                        RNG &seeding_rng = GlobalRng::rng);
    DeviceBregVsSampler(RegressionModel *model, double prior_sigma_nobs,
                        double prior_sigma_guess, double prior_beta_nobs,
                        double diagonal_shrinkage, double prior_inclusion_probability,
                        bool force_intercept,
                        int chains, int device = 0, int lookahead = 256,
                        RNG &seeding_rng = GlobalRng::rng);
    // #3 (:144-160) and #4 (:162-180): the prior's parameters, as numbers or as a struct
    DeviceBregVsSampler(RegressionModel *model, const Vector &prior_mean,
                        const SpdMatrix &unscaled_prior_precision, double sigma_guess,
                        double df, const Vector &prior_inclusion_probs,
                        int chains, int device = 0, int lookahead = 256,
                        RNG &seeding_rng = GlobalRng::rng);
    DeviceBregVsSampler(RegressionModel *model, const ZellnerPriorParameters &prior,
                        int chains, int device = 0, int lookahead = 256,
                        RNG &seeding_rng = GlobalRng::rng);
    // #5 (:182-194): the prior objects themselves.  "If external copies of the pointers
    // supplied to the constructor are kept then the values of the prior parameters can be
    // modified" (BregVsSampler.hpp:98-101): the sampler observes the three models'
    // parameters (Data::add_observer, DataTypes.hpp:76) and the next draw() uploads what
    // changed before it launches.  (What a model keeps outside its Params -- the slab's
    // unscaled precision -- has no signal: call refresh_priors() after changing it.)
    DeviceBregVsSampler(RegressionModel *model,
                        const Ptr<MvnGivenScalarSigmaBase> &slab,
                        const Ptr<GammaModelBase> &residual_precision_prior,
                        const Ptr<VariableSelectionPrior> &spike,
                        int chains, int device = 0, int lookahead = 256,
                        RNG &seeding_rng = GlobalRng::rng);
    // The same sampler over several devices of one node (ba_group_*): engine i on
    // devices[i] owns the global chains [i * chains_per_device, (i + 1) *
    // chains_per_device), so the draws of a chain do not depend on the device list.
    // Chain 0 (the one the model sees) lives on devices[0].
    DeviceBregVsSampler(RegressionModel *model,
                        const Ptr<MvnGivenScalarSigmaBase> &slab,
                        const Ptr<GammaModelBase> &residual_precision_prior,
                        const Ptr<VariableSelectionPrior> &spike,
                        int chains_per_device, const std::vector<int> &devices,
                        int lookahead = 256, RNG &seeding_rng = GlobalRng::rng);
    ~DeviceBregVsSampler() override;

    void draw() override;            // BregVsSampler::draw, BregVsSampler.cpp:252-261
    double logpri() const override;  // BregVsSampler::logpri, :380-393 (chain 0)

    // PosteriorSampler::set_seed is not virtual in BOOM; the engine's streams
    // are re-keyed through this one
    void set_device_seed(unsigned long seed);
    unsigned long device_seed() const { return device_seed_; }

    // BregVsSampler's setters, same names and meaning (BregVsSampler.hpp:131-161)
    void limit_model_selection(uint max_flips);
    void suppress_model_selection();
    void allow_model_selection();
    void set_correlation_swap_threshold(double threshold);
    void set_sigma_upper_limit(double sigma_upper_limit);

    // the prior objects (BregVsSampler has no accessors for them; kept for the callers of
    // constructors #1 - #4, who have no pointers of their own)
    const Ptr<MvnGivenScalarSigmaBase> &slab() const { return slab_; }
    const Ptr<GammaModelBase> &residual_precision_prior() const { return residual_precision_prior_; }
    const Ptr<VariableSelectionPrior> &spike() const { return spike_; }
    // the device copies of the priors are stale: the next draw() reads the objects again
    void refresh_priors() { priors_stale_ = true; }

    // new: how many draws a launch runs ahead of the caller (ba_set_lookahead)
    void set_lookahead(int n);
    // new: the other chains
    int number_of_chains() const { return chains_; }
    int number_of_devices() const { return static_cast<int>(engines_.size()); }
    void chain_state(int chain, Selector &inc, Vector &beta, double &sigsq) const;

   private:
    void check(int rc) const;
    void create_engine(int device, RNG &seeding_rng);   // the single-device constructors' first step
    void destroy_engines();
    void upload_suf();
    void priors_from_engine();          // (#1, #2) the objects from what ba_set_priors_ctor* assembled
    void configure(int lookahead);      // the objects -> every engine, observers, state, look-ahead
    void upload_priors();
    void observe();
    void unobserve();
    void options();
    void push_state();   // coef().inc(), Beta(), sigsq() -> every chain
    void pull_chain0();  // chain 0 -> coef().set_inc / set_Beta / set_sigsq
    RegressionModel *model_;
    ba_group *group_;                 // null for the single-device constructor
    std::vector<ba_engine *> engines_;  // owned by group_ when there is one
    int chains_;
    unsigned long device_seed_;
    int max_flips_;
    double swap_threshold_;
    double prior_df_, prior_sigma_guess_;
    double sigma_upper_limit_;
    Ptr<MvnGivenScalarSigmaBase> slab_;
    Ptr<GammaModelBase> residual_precision_prior_;
    Ptr<VariableSelectionPrior> spike_;
    bool priors_stale_;
  };

}  // namespace BOOM
#endif  // BOOM_AMD_DEVICE_BREG_VS_SAMPLER_HPP_
