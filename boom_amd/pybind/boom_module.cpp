// pybind11 module `boom_amd._boom`: the slice of BayesBoom's `boom` module that
// sits on the spike-and-slab path, with the same names and call shapes as the
// reference's bindings (Interfaces/python/BayesBoom/Models/Glm/GlmModel_def.cpp:
// 787-828 BregVsSampler; RegressionModel, GlmCoefs; Models/Model_def.cpp
// MvnGivenScalarSigma / ChisqModel / VariableSelectionPrior;
// ModelWrapper.cpp:89-119 sample_posterior / set_method), so that the driver
// loop of Interfaces/python/spikeslab/BayesBoom/spikeslab/spikeslab.py:191-207
//
//     model = boom.RegressionModel(X, y, False)
//     sampler = boom.BregVsSampler(model, slab, siginv_prior, spike)
//     model.set_method(sampler)
//     model.coef.drop_all(); model.coef.add(0)
//     for i in range(niter):
//         model.sample_posterior(); record(model.sigma, model.coef)
//
// runs unchanged with `import boom_amd._boom as boom`.  Everything goes through
// the C++ host side (include/boom_amd.hpp) and from there the C-ABI: no compute
// happens in this file.  Extra keyword arguments (chains=, seed=, device=) and
// accessors (chain_states, set_lookahead) expose what is new: many chains.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "boom_amd.hpp"

namespace py = pybind11;
using namespace boom_amd_api;

namespace {

typedef py::array_t<double, py::array::c_style | py::array::forcecast> NpArray;

Matrix matrix_from(const NpArray &a) {
  if (a.ndim() != 2) throw std::runtime_error("expected a 2-d array");
  Matrix m((int)a.shape(0), (int)a.shape(1));
  auto r = a.unchecked<2>();
  for (py::ssize_t i = 0; i < a.shape(0); ++i)
    for (py::ssize_t j = 0; j < a.shape(1); ++j) m((int)i, (int)j) = r(i, j);
  return m;
}
Vector vector_from(const NpArray &a) {
  // (c_style | forcecast: the data are contiguous doubles whatever the shape)
  const double *d = a.data();
  return Vector(d, d + a.size());
}
py::array_t<double> to_numpy(const Vector &v) {
  py::array_t<double> out((py::ssize_t)v.size());
  auto w = out.mutable_unchecked<1>();
  for (size_t i = 0; i < v.size(); ++i) w((py::ssize_t)i) = v[i];
  return out;
}

// GlmCoefs as the drivers use it: model.coef.drop_all() / add / inc / Beta
struct CoefView {
  RegressionModel *model;
};

// boom.BinomialLogitModel(xdim, include_all) / BinomialProbitModel, filled by
// add_dataset(successes, trials, predictors) as in BayesBoom (GlmModel_def.cpp:549-590).
// The reference's sampler reads the model's data at every draw; here the data go to
// the device once, when the sampler -- which brings clt_threshold -- is attached.
struct PyBinomialModel {
  int xdim;
  bool logit;
  int chains;
  uint64_t seed;
  int device;
  std::vector<uint8_t> inc;          // coef.inc until the device model exists
  Matrix X;
  Vector y, n;
  bool have_data = false;
  Ptr<BinomialRegressionModelBase> impl;

  PyBinomialModel(int xdim_, bool include_all, bool logit_, int chains_, uint64_t seed_, int device_)
      : xdim(xdim_), logit(logit_), chains(chains_), seed(seed_), device(device_),
        inc((size_t)xdim_, include_all ? 1 : 0) {
    if (xdim_ <= 0) throw std::runtime_error("xdim must be positive");
    if (!include_all) inc[0] = 1;    // "only the intercept starts out included"
  }
  void materialise(int clt_threshold) {
    if (!have_data) throw std::runtime_error("add_dataset(successes, trials, predictors) before attaching a sampler");
    // a sampler holds a raw pointer to the device model (as BOOM's samplers hold their
    // model): replacing it under a sampler already attached would leave that sampler
    // dangling, so the device model is created exactly once
    if (impl) throw std::runtime_error("a sampler has already been attached to this model: create a new model to change clt_threshold or the sampler");
    if (logit) impl.reset(new BinomialLogitModel(X, y, n, clt_threshold, chains, seed, device));
    else impl.reset(new BinomialProbitModel(X, y, n, clt_threshold, chains, seed, device));
    impl->drop_all();
    for (int j = 0; j < xdim; ++j)
      if (inc[j]) impl->add(j);
  }
  void need_impl() const {
    if (!impl) throw std::runtime_error("no sampler attached to the model yet");
  }
};
struct BinomialCoefView {
  PyBinomialModel *model;
};

template <class SamplerT, class ModelT>
Ptr<SamplerT> make_binomial_sampler(PyBinomialModel &m, const Ptr<MvnModel> &slab,
                                    const Ptr<VariableSelectionPrior> &spike, int clt_threshold) {
  m.materialise(clt_threshold);
  return Ptr<SamplerT>(new SamplerT(static_cast<ModelT *>(m.impl.get()), slab, spike));
}

template <class PyClass>
void bind_binomial_model(PyClass &c) {
  c.def("add_dataset",
        [](PyBinomialModel &m, const NpArray &successes, const NpArray &trials, const NpArray &predictors) {
          m.X = matrix_from(predictors);
          m.y = vector_from(successes);
          m.n = vector_from(trials);
          if (m.X.ncol() != m.xdim) throw std::runtime_error("predictors do not match xdim");
          m.have_data = true;
        },
        py::arg("successes"), py::arg("trials"), py::arg("predictors"))
      .def_property_readonly("xdim", [](const PyBinomialModel &m) { return m.xdim; })
      .def_property_readonly("coef", py::cpp_function([](PyBinomialModel &m) { return BinomialCoefView{&m}; },
                                                      py::keep_alive<0, 1>()))
      .def_property_readonly("Beta", [](const PyBinomialModel &m) { m.need_impl(); return to_numpy(m.impl->Beta()); })
      .def("set_method", [](PyBinomialModel &m, const Ptr<PosteriorSampler> &s) { m.need_impl(); m.impl->set_method(s); })
      .def("sample_posterior", [](PyBinomialModel &m) { m.need_impl(); m.impl->sample_posterior(); })
      .def("chain_states", [](const PyBinomialModel &m) {
        m.need_impl();
        std::vector<uint8_t> g;
        Vector b;
        m.impl->chain_states(g, b);
        const py::ssize_t p = m.xdim, C = (py::ssize_t)g.size() / p;
        py::array_t<uint8_t> G({C, p});
        py::array_t<double> B({C, p});
        std::memcpy(G.mutable_data(), g.data(), g.size());
        std::memcpy(B.mutable_data(), b.data(), b.size() * 8);
        return py::make_tuple(G, B);
      }, "inclusion indicators and coefficients of EVERY chain");
}

}  // namespace

PYBIND11_MODULE(_boom, boom) {
  boom.doc() = "BayesBoom-shaped bindings of the MI355X spike-and-slab engine (boom_amd)";

  py::class_<PosteriorSampler, Ptr<PosteriorSampler>>(boom, "PosteriorSampler")
      .def("draw", &PosteriorSampler::draw)
      .def("logpri", &PosteriorSampler::logpri)
      .def("set_seed", &PosteriorSampler::set_seed);

  py::class_<MvnGivenScalarSigma, Ptr<MvnGivenScalarSigma>>(boom, "MvnGivenScalarSigma")
      .def(py::init([](const NpArray &mean, const NpArray &unscaled_precision, py::object) {
             return new MvnGivenScalarSigma(vector_from(mean), matrix_from(unscaled_precision));
           }),
           py::arg("mean"), py::arg("unscaled_precision"), py::arg("sigsq") = py::none(),
           "The slab: beta | sigma ~ N(mean, sigma^2 unscaled_precision^{-1}).")
      .def_property_readonly("dim", &MvnGivenScalarSigma::dim);

  py::class_<ChisqModel, Ptr<ChisqModel>>(boom, "ChisqModel")
      .def(py::init<double, double>(), py::arg("df"), py::arg("sigma_estimate"))
      .def_property_readonly("df", &ChisqModel::df)
      .def_property_readonly("sigma", &ChisqModel::sigma);

  py::class_<VariableSelectionPrior, Ptr<VariableSelectionPrior>>(boom, "VariableSelectionPrior")
      .def(py::init([](const NpArray &probs) { return new VariableSelectionPrior(vector_from(probs)); }),
           py::arg("prior_inclusion_probabilities"))
      .def("set_max_model_size", &VariableSelectionPrior::set_max_model_size)
      .def_property_readonly("potential_nvars", &VariableSelectionPrior::potential_nvars);

  py::class_<CoefView>(boom, "GlmCoefs")
      .def("drop_all", [](CoefView &c) { c.model->drop_all(); })
      .def("add", [](CoefView &c, int i) { c.model->add(i); })
      .def("drop", [](CoefView &c, int i) { c.model->drop(i); })
      .def_property_readonly("inc", [](const CoefView &c) {
        const RegressionModel &m = *c.model;
        std::vector<bool> g(m.xdim());
        for (int j = 0; j < m.xdim(); ++j) g[j] = m.inc()[j];
        return g;
      })
      .def_property_readonly("nvars", [](const CoefView &c) {
        const RegressionModel &m = *c.model;
        return m.inc().nvars();
      })
      .def_property_readonly("Beta", [](const CoefView &c) {
        const RegressionModel &m = *c.model;
        return to_numpy(m.Beta());
      });

  py::class_<RegressionModel, Ptr<RegressionModel>>(boom, "RegressionModel")
      .def(py::init([](const NpArray &X, const NpArray &y, bool, int chains, uint64_t seed, int device,
                       const std::vector<int> &devices) {
             if (!devices.empty())
               return new RegressionModel(matrix_from(X), vector_from(y), chains, devices, seed);
             return new RegressionModel(matrix_from(X), vector_from(y), chains, seed, device);
           }),
           py::arg("X"), py::arg("y"), py::arg("start_at_mle") = false, py::arg("chains") = 1,
           py::arg("seed") = 8675309ull, py::arg("device") = 0, py::arg("devices") = std::vector<int>(),
           "RegressionModel(X, y, start_at_mle): sufficient statistics are built on the "
           "device.  chains / seed / device: the many-chain engine behind the model; "
           "devices=[...]: `chains` chains on EACH listed device behind the one model.")
      .def_property_readonly("xdim", &RegressionModel::xdim)
      .def_property_readonly("coef", py::cpp_function([](RegressionModel &m) { return CoefView{&m}; },
                                                      py::keep_alive<0, 1>()))
      .def_property_readonly("Beta", [](const RegressionModel &m) { return to_numpy(m.Beta()); })
      .def_property_readonly("sigsq", &RegressionModel::sigsq)
      .def_property_readonly("sigma", [](const RegressionModel &m) { return std::sqrt(m.sigsq()); })
      .def("set_sigsq", &RegressionModel::set_sigsq)
      .def("set_method", [](RegressionModel &m, const Ptr<PosteriorSampler> &s) { m.set_method(s); })
      .def("clear_methods", &RegressionModel::clear_methods)
      .def("sample_posterior", &RegressionModel::sample_posterior)
      .def("chain_states", [](const RegressionModel &m) {
        std::vector<uint8_t> g;
        Vector b, s;
        m.chain_states(g, b, s);
        const py::ssize_t C = (py::ssize_t)s.size(), p = m.xdim();
        py::array_t<uint8_t> G({C, p});
        py::array_t<double> B({C, p});
        std::memcpy(G.mutable_data(), g.data(), g.size());
        std::memcpy(B.mutable_data(), b.data(), b.size() * 8);
        return py::make_tuple(G, B, to_numpy(s));
      }, "inclusion indicators, coefficients and sigma^2 of EVERY chain");

  py::class_<BregVsSampler, PosteriorSampler, Ptr<BregVsSampler>>(boom, "BregVsSampler")
      .def(py::init([](RegressionModel *model, const Ptr<MvnGivenScalarSigma> &slab,
                       const Ptr<ChisqModel> &residual_precision_prior,
                       const Ptr<VariableSelectionPrior> &spike, py::object) {
             return new BregVsSampler(model, slab, residual_precision_prior, spike);
           }),
           py::arg("model"), py::arg("slab"), py::arg("residual_precision_prior"), py::arg("spike"),
           py::arg("seeding_rng") = py::none(), py::keep_alive<1, 2>(),
           "Create a BregVsSampler -- a spike and slab sampler for regression models.")
      .def("limit_model_selection", [](BregVsSampler &s, int max_flips) { s.limit_model_selection((uint)max_flips); })
      .def("suppress_model_selection", &BregVsSampler::suppress_model_selection)
      .def("set_sigma_upper_limit", &BregVsSampler::set_sigma_upper_limit)
      .def("set_correlation_swap_threshold", &BregVsSampler::set_correlation_swap_threshold)
      .def("set_lookahead", &BregVsSampler::set_lookahead,
           "run n sweeps per launch and hand them out one sample_posterior() at a time");

  // ---- logit / probit spike and slab (GlmModel_def.cpp:549-590, :966-1010) ----------
  py::class_<MvnModel, Ptr<MvnModel>>(boom, "MvnModel")
      .def(py::init([](const NpArray &mu, const NpArray &Sigma, bool ivar) {
             if (!ivar) throw std::runtime_error("MvnModel: pass the precision (ivar=True); the engine works in precisions");
             return new MvnModel(vector_from(mu), matrix_from(Sigma));
           }),
           py::arg("mu"), py::arg("Sigma"), py::arg("ivar") = false,
           "A slab whose precision does not scale with sigma^2 (MvnBase).")
      .def_property_readonly("dim", &MvnModel::dim);

  py::class_<BinomialCoefView>(boom, "BinomialGlmCoefs")
      .def("drop_all", [](BinomialCoefView &c) {
        std::fill(c.model->inc.begin(), c.model->inc.end(), 0);
        if (c.model->impl) c.model->impl->drop_all();
      })
      .def("add", [](BinomialCoefView &c, int i) {
        c.model->inc.at(i) = 1;
        if (c.model->impl) c.model->impl->add(i);
      })
      .def("drop", [](BinomialCoefView &c, int i) {
        c.model->inc.at(i) = 0;
        if (c.model->impl) c.model->impl->drop(i);
      })
      .def_property_readonly("inc", [](const BinomialCoefView &c) {
        std::vector<bool> g(c.model->xdim);
        for (int j = 0; j < c.model->xdim; ++j) g[j] = c.model->impl ? c.model->impl->inc()[j] : c.model->inc[j] != 0;
        return g;
      })
      .def_property_readonly("Beta", [](const BinomialCoefView &c) {
        c.model->need_impl();
        return to_numpy(c.model->impl->Beta());
      });

  struct PyLogit : PyBinomialModel { using PyBinomialModel::PyBinomialModel; };
  struct PyProbit : PyBinomialModel { using PyBinomialModel::PyBinomialModel; };
  py::class_<PyBinomialModel>(boom, "_BinomialRegressionModel");
  py::class_<PyLogit, PyBinomialModel> logit(boom, "BinomialLogitModel");
  logit.def(py::init([](int xdim, bool include_all, int chains, uint64_t seed, int device) {
              return new PyLogit(xdim, include_all, true, chains, seed, device);
            }),
            py::arg("xdim"), py::arg("include_all") = true, py::arg("chains") = 1,
            py::arg("seed") = 8675309ull, py::arg("device") = 0);
  bind_binomial_model(logit);
  py::class_<PyProbit, PyBinomialModel> probit(boom, "BinomialProbitModel");
  probit.def(py::init([](int xdim, bool include_all, int chains, uint64_t seed, int device) {
               return new PyProbit(xdim, include_all, false, chains, seed, device);
             }),
             py::arg("xdim"), py::arg("include_all") = true, py::arg("chains") = 1,
             py::arg("seed") = 8675309ull, py::arg("device") = 0);
  bind_binomial_model(probit);

  py::class_<BinomialLogitSpikeSlabSampler, PosteriorSampler, Ptr<BinomialLogitSpikeSlabSampler>>(
      boom, "BinomialLogitSpikeSlabSampler")
      .def(py::init([](PyLogit &model, const Ptr<MvnModel> &slab, const Ptr<VariableSelectionPrior> &spike,
                       int clt_threshold, py::object) {
             return make_binomial_sampler<BinomialLogitSpikeSlabSampler, BinomialLogitModel>(model, slab, spike,
                                                                                             clt_threshold);
           }),
           py::arg("model"), py::arg("slab"), py::arg("spike"), py::arg("clt_threshold") = 5,
           py::arg("seeding_rng") = py::none(), py::keep_alive<1, 2>())
      .def("limit_model_selection", &BinomialLogitSpikeSlabSampler::limit_model_selection);
  py::class_<BinomialProbitSpikeSlabSampler, PosteriorSampler, Ptr<BinomialProbitSpikeSlabSampler>>(
      boom, "BinomialProbitSpikeSlabSampler")
      .def(py::init([](PyProbit &model, const Ptr<MvnModel> &slab, const Ptr<VariableSelectionPrior> &spike,
                       int clt_threshold, py::object) {
             return make_binomial_sampler<BinomialProbitSpikeSlabSampler, BinomialProbitModel>(model, slab, spike,
                                                                                               clt_threshold);
           }),
           py::arg("model"), py::arg("slab"), py::arg("spike"), py::arg("clt_threshold") = 5,
           py::arg("seeding_rng") = py::none(), py::keep_alive<1, 2>())
      .def("limit_model_selection", &BinomialProbitSpikeSlabSampler::limit_model_selection);

  // ---- bsts: state models, the state-space regression and its sampler
  // (StateModelWrapper.cpp:73-260, StateSpaceModelWrapper.cpp:203-260, :476-490).  The
  // reference attaches one sampler object per variance and one to the observation
  // model; here each state model takes its prior through set_prior and the
  // observation model's three priors go to StateSpacePosteriorSampler.
  py::class_<LocalLevelStateModel, Ptr<LocalLevelStateModel>>(boom, "LocalLevelStateModel")
      .def(py::init<double>(), py::arg("sigma") = 1.0)
      .def("set_initial_state_mean", &LocalLevelStateModel::set_initial_state_mean)
      .def("set_initial_state_variance", &LocalLevelStateModel::set_initial_state_variance)
      .def("set_prior", &LocalLevelStateModel::set_prior, py::arg("df"), py::arg("sigma_guess"),
           py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity(),
           "ZeroMeanGaussianConjSampler(model, df, sigma_guess) + set_sigma_upper_limit");
  py::class_<LocalLinearTrendStateModel, Ptr<LocalLinearTrendStateModel>>(boom, "LocalLinearTrendStateModel")
      .def(py::init<>())
      .def("set_initial_state_mean", [](LocalLinearTrendStateModel &m, const NpArray &v) { m.set_initial_state_mean(vector_from(v)); })
      .def("set_initial_state_variance", [](LocalLinearTrendStateModel &m, const NpArray &diag) { m.set_initial_state_variance(vector_from(diag)); })
      .def("set_initial_sigma", &LocalLinearTrendStateModel::set_initial_sigma)
      .def("set_prior", &LocalLinearTrendStateModel::set_prior, py::arg("which_variable"), py::arg("df"),
           py::arg("sigma_guess"), py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity(),
           "ZeroMeanMvnIndependenceSampler(model, df, sigma_guess, which_variable) + set_sigma_upper_limit");
  py::class_<SeasonalStateModel, Ptr<SeasonalStateModel>>(boom, "SeasonalStateModel")
      .def(py::init<int, int>(), py::arg("nseasons"), py::arg("season_duration") = 1)
      .def_property_readonly("state_dimension", &SeasonalStateModel::state_dimension)
      .def_property_readonly("nseasons", &SeasonalStateModel::nseasons)
      .def_property_readonly("season_duration", &SeasonalStateModel::season_duration)
      .def("set_time_of_first_observation", &SeasonalStateModel::set_time_of_first_observation)
      .def("set_sigsq", &SeasonalStateModel::set_sigsq)
      .def("set_initial_state_mean", [](SeasonalStateModel &m, const NpArray &v) { m.set_initial_state_mean(vector_from(v)); })
      .def("set_initial_state_variance", &SeasonalStateModel::set_initial_state_variance)
      .def("set_prior", &SeasonalStateModel::set_prior, py::arg("df"), py::arg("sigma_guess"),
           py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity());

  py::class_<ArStateModel, Ptr<ArStateModel>>(boom, "ArStateModel")
      .def(py::init<int>(), py::arg("number_of_lags"))
      .def_property_readonly("state_dimension", &ArStateModel::state_dimension)
      .def_property_readonly("number_of_lags", &ArStateModel::number_of_lags)
      .def("set_phi", [](ArStateModel &m, const NpArray &v) { m.set_phi(vector_from(v)); })
      .def("set_sigma", &ArStateModel::set_sigma)
      .def("set_sigsq", &ArStateModel::set_sigsq)
      .def("set_initial_state_mean", [](ArStateModel &m, const NpArray &v) { m.set_initial_state_mean(vector_from(v)); })
      .def("set_initial_state_variance", &ArStateModel::set_initial_state_variance)
      .def("set_prior", &ArStateModel::set_prior, py::arg("df"), py::arg("sigma_guess"),
           py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity(),
           "ArPosteriorSampler(model, ChisqModel(df, sigma_guess)) + set_sigma_upper_limit");

  py::class_<StaticInterceptStateModel, Ptr<StaticInterceptStateModel>>(boom, "StaticInterceptStateModel")
      .def(py::init<>())
      .def_property_readonly("state_dimension", &StaticInterceptStateModel::state_dimension)
      .def("set_initial_state_mean", &StaticInterceptStateModel::set_initial_state_mean)
      .def("set_initial_state_variance", &StaticInterceptStateModel::set_initial_state_variance);

  py::class_<TrigStateModel, Ptr<TrigStateModel>>(boom, "TrigStateModel")
      .def(py::init([](double period, const NpArray &frequencies) {
             return new TrigStateModel(period, vector_from(frequencies));
           }),
           py::arg("period"), py::arg("frequencies"))
      .def_property_readonly("state_dimension", &TrigStateModel::state_dimension)
      .def("set_sigsq", &TrigStateModel::set_sigsq)
      .def("set_initial_state_mean", [](TrigStateModel &m, const NpArray &v) { m.set_initial_state_mean(vector_from(v)); })
      .def("set_initial_state_variance",
           [](TrigStateModel &m, const NpArray &v) { m.set_initial_state_variance(vector_from(v)); },
           "the diagonal of the initial state's variance")
      .def("set_prior", &TrigStateModel::set_prior, py::arg("df"), py::arg("sigma_guess"),
           py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity(),
           "ZeroMeanGaussianConjSampler(error_distribution(), ChisqModel(df, sigma_guess)) + set_sigma_upper_limit");

  py::class_<ZeroMeanGaussianModel, Ptr<ZeroMeanGaussianModel>>(boom, "ZeroMeanGaussianModel")
      .def(py::init<double>(), py::arg("sigma") = 1.0)
      .def_property_readonly("sigma", &ZeroMeanGaussianModel::sigma);
  py::class_<NonzeroMeanAr1Model, Ptr<NonzeroMeanAr1Model>>(boom, "NonzeroMeanAr1Model")
      .def(py::init<double, double, double>(), py::arg("mu") = 0.0, py::arg("phi") = 0.0, py::arg("sigma") = 1.0)
      .def_property_readonly("mu", &NonzeroMeanAr1Model::mu)
      .def_property_readonly("phi", &NonzeroMeanAr1Model::phi)
      .def_property_readonly("sigma", &NonzeroMeanAr1Model::sigma);
  py::class_<SemilocalLinearTrendStateModel, Ptr<SemilocalLinearTrendStateModel>>(boom, "SemilocalLinearTrendStateModel")
      .def(py::init<const Ptr<ZeroMeanGaussianModel> &, const Ptr<NonzeroMeanAr1Model> &>(), py::arg("level"),
           py::arg("slope"))
      .def_property_readonly("state_dimension", &SemilocalLinearTrendStateModel::state_dimension)
      .def("set_initial_level_mean", &SemilocalLinearTrendStateModel::set_initial_level_mean)
      .def("set_initial_level_sd", &SemilocalLinearTrendStateModel::set_initial_level_sd)
      .def("set_initial_slope_mean", &SemilocalLinearTrendStateModel::set_initial_slope_mean)
      .def("set_initial_slope_sd", &SemilocalLinearTrendStateModel::set_initial_slope_sd)
      .def("set_level_prior", &SemilocalLinearTrendStateModel::set_level_prior, py::arg("df"), py::arg("sigma_guess"),
           py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity(),
           "ZeroMeanGaussianConjSampler(level, df, sigma_guess) + set_sigma_upper_limit")
      .def("set_slope_prior", &SemilocalLinearTrendStateModel::set_slope_prior, py::arg("mean_mu"),
           py::arg("mean_sigma"), py::arg("ar1_mu"), py::arg("ar1_sigma"), py::arg("df"), py::arg("sigma_guess"),
           py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity(), py::arg("force_stationary") = true,
           py::arg("force_ar1_positive") = false,
           "NonzeroMeanAr1Sampler(slope, mean prior, AR(1) coefficient prior, ChisqModel(df, sigma_guess)) + its switches");

  py::class_<StateSpaceRegressionModel, Ptr<StateSpaceRegressionModel>>(boom, "StateSpaceRegressionModel")
      .def(py::init([](const NpArray &response, const NpArray &predictors, const std::vector<bool> &is_observed,
                       int chains, uint64_t seed, int device) {
             return new StateSpaceRegressionModel(vector_from(response), matrix_from(predictors), is_observed,
                                                  chains, seed, device);
           }),
           py::arg("response"), py::arg("predictors"), py::arg("is_observed") = std::vector<bool>(),
           py::arg("chains") = 1, py::arg("seed") = 8675309ull, py::arg("device") = 0)
      .def_property_readonly("xdim", &StateSpaceRegressionModel::xdim)
      .def_property_readonly("time_dimension", &StateSpaceRegressionModel::time_dimension)
      .def_property_readonly("state_dimension", &StateSpaceRegressionModel::state_dimension)
      .def("add_state", [](StateSpaceRegressionModel &m, const Ptr<LocalLevelStateModel> &s) { m.add_state(s); })
      .def("add_state", [](StateSpaceRegressionModel &m, const Ptr<LocalLinearTrendStateModel> &s) { m.add_state(s); })
      .def("add_state", [](StateSpaceRegressionModel &m, const Ptr<SeasonalStateModel> &s) { m.add_state(s); })
      .def("add_state", [](StateSpaceRegressionModel &m, const Ptr<ArStateModel> &s) { m.add_state(s); })
      .def("add_state", [](StateSpaceRegressionModel &m, const Ptr<StaticInterceptStateModel> &s) { m.add_state(s); })
      .def("add_state", [](StateSpaceRegressionModel &m, const Ptr<TrigStateModel> &s) { m.add_state(s); })
      .def("add_state", [](StateSpaceRegressionModel &m, const Ptr<SemilocalLinearTrendStateModel> &s) { m.add_state(s); })
      .def("semilocal_slope", [](const StateSpaceRegressionModel &m, int chain, int which) { return to_numpy(m.semilocal_slope(chain, which)); },
           py::arg("chain") = 0, py::arg("which") = 0,
           "(AR(1) coefficient, long-run mean) of the which-th SemilocalLinearTrendStateModel's slope in one chain's draw")
      .def_property_readonly("number_of_state_models", &StateSpaceRegressionModel::number_of_state_models)
      .def("ar_phi", [](const StateSpaceRegressionModel &m, int chain, int which) { return to_numpy(m.ar_phi(chain, which)); },
           py::arg("chain") = 0, py::arg("which") = 0,
           "the coefficients of the which-th ArStateModel in one chain's current draw")
      .def("ar_sigsq", &StateSpaceRegressionModel::ar_sigsq, py::arg("chain") = 0, py::arg("which") = 0)
      .def("set_method", [](StateSpaceRegressionModel &m, const Ptr<PosteriorSampler> &s) { m.set_method(s); })
      .def("sample_posterior", &StateSpaceRegressionModel::sample_posterior)
      .def("state", [](const StateSpaceRegressionModel &m, int chain) {
        if (!m.structural()) return py::array(to_numpy(m.state(chain)));
        const Matrix st = m.structural_state(chain);
        py::array_t<double> out({(py::ssize_t)st.nrow(), (py::ssize_t)st.ncol()});
        auto w = out.mutable_unchecked<2>();
        for (int i = 0; i < st.nrow(); ++i)
          for (int j = 0; j < st.ncol(); ++j) w(i, j) = st(i, j);
        return py::array(out);
      }, py::arg("chain") = 0, "the state draw of one chain: (T,) for the local level, (state_dimension, T) otherwise")
      .def("state_variances", [](const StateSpaceRegressionModel &m, int chain) {
        if (!m.structural()) { Vector v(1, m.level_sigsq(chain)); return to_numpy(v); }
        return to_numpy(m.state_variances(chain));
      }, py::arg("chain") = 0, "every variance parameter in state-model order (a local linear trend has two)")
      .def("chain_states", [](const StateSpaceRegressionModel &m) {
        const py::ssize_t C = m.engine()->chains(), p = m.xdim();
        py::array_t<uint8_t> G({C, p});
        py::array_t<double> B({C, p}), S((py::ssize_t)C);
        m.engine()->check(ba_get_states(m.engine()->get(), G.mutable_data(), B.mutable_data(), S.mutable_data()));
        return py::make_tuple(G, B, S);
      }, "inclusion indicators, coefficients and residual variances of EVERY chain");

  py::class_<StateSpacePosteriorSampler, PosteriorSampler, Ptr<StateSpacePosteriorSampler>>(
      boom, "StateSpacePosteriorSampler")
      .def(py::init([](StateSpaceRegressionModel *model, const Ptr<MvnGivenScalarSigma> &slab,
                       const Ptr<ChisqModel> &residual_precision_prior, const Ptr<VariableSelectionPrior> &spike,
                       double sigma_upper_limit, py::object) {
             return new StateSpacePosteriorSampler(model, slab, residual_precision_prior, spike, sigma_upper_limit);
           }),
           py::arg("model"), py::arg("slab"), py::arg("residual_precision_prior"), py::arg("spike"),
           py::arg("sigma_upper_limit") = std::numeric_limits<double>::infinity(),
           py::arg("seeding_rng") = py::none(), py::keep_alive<1, 2>())
      .def("set_lookahead", &StateSpacePosteriorSampler::set_lookahead,
           "enqueue n sweep rounds at a time and hand them out one sample_posterior() at a time (default 64)");

  // ---- Poisson regression spike and slab (GlmModel_def.cpp: PoissonRegressionModel,
  // PoissonRegressionSpikeSlabSampler) ---------------------------------------------------
  py::class_<PoissonRegressionModel, Ptr<PoissonRegressionModel>>(boom, "PoissonRegressionModel")
      .def(py::init([](const NpArray &X, const NpArray &y, const NpArray &exposure, int chains, uint64_t seed,
                       int device) {
             return new PoissonRegressionModel(matrix_from(X), vector_from(y), vector_from(exposure), chains, seed,
                                               device);
           }),
           py::arg("predictors"), py::arg("response"), py::arg("exposure"), py::arg("chains") = 1,
           py::arg("seed") = 8675309ull, py::arg("device") = 0)
      .def_property_readonly("xdim", &PoissonRegressionModel::xdim)
      .def("set_mixture_table", [](PoissonRegressionModel &m, const std::vector<int64_t> &counts,
                                   const std::vector<int32_t> &ncomp, const NpArray &mu, const NpArray &sigma,
                                   const NpArray &weight, int64_t largest_index) {
             NormalMixtureTable t;
             t.counts = counts; t.ncomp = ncomp;
             t.mu = vector_from(mu); t.sigma = vector_from(sigma); t.weight = vector_from(weight);
             t.largest_index = largest_index;
             m.set_mixture_table(t);
           },
           py::arg("counts"), py::arg("ncomp"), py::arg("mu"), py::arg("sigma"), py::arg("weight"),
           py::arg("largest_index"),
           "the reference's normal mixtures for the negative log-gamma densities (its data)")
      .def("drop_all", &PoissonRegressionModel::drop_all)
      .def("add", &PoissonRegressionModel::add)
      .def("drop", &PoissonRegressionModel::drop)
      .def_property_readonly("inc", [](const PoissonRegressionModel &m) {
        std::vector<bool> g(m.xdim());
        for (int j = 0; j < m.xdim(); ++j) g[j] = m.inc()[j];
        return g;
      })
      .def_property_readonly("Beta", [](const PoissonRegressionModel &m) { return to_numpy(m.Beta()); })
      .def("set_Beta", [](PoissonRegressionModel &m, const NpArray &b) { m.set_Beta(vector_from(b)); })
      .def("set_method", [](PoissonRegressionModel &m, const Ptr<PosteriorSampler> &s) { m.set_method(s); })
      .def("sample_posterior", &PoissonRegressionModel::sample_posterior)
      .def("chain_states", [](const PoissonRegressionModel &m) {
        const py::ssize_t C = m.engine()->chains(), p = m.xdim();
        py::array_t<uint8_t> G({C, p});
        py::array_t<double> B({C, p});
        m.engine()->check(ba_get_states(m.engine()->get(), G.mutable_data(), B.mutable_data(), nullptr));
        return py::make_tuple(G, B);
      }, "inclusion indicators and coefficients of EVERY chain");
  py::class_<PoissonRegressionSpikeSlabSampler, PosteriorSampler, Ptr<PoissonRegressionSpikeSlabSampler>>(
      boom, "PoissonRegressionSpikeSlabSampler")
      .def(py::init([](PoissonRegressionModel *model, const Ptr<MvnModel> &slab,
                       const Ptr<VariableSelectionPrior> &spike, int, py::object) {
             return new PoissonRegressionSpikeSlabSampler(model, slab, spike);
           }),
           py::arg("model"), py::arg("slab"), py::arg("spike"), py::arg("number_of_threads") = 1,
           py::arg("seeding_rng") = py::none(), py::keep_alive<1, 2>())
      .def("limit_model_selection", &PoissonRegressionSpikeSlabSampler::limit_model_selection);
}
