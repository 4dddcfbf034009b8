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
      .def(py::init([](const NpArray &X, const NpArray &y, bool, int chains, uint64_t seed, int device) {
             return new RegressionModel(matrix_from(X), vector_from(y), chains, seed, device);
           }),
           py::arg("X"), py::arg("y"), py::arg("start_at_mle") = false, py::arg("chains") = 1,
           py::arg("seed") = 8675309ull, py::arg("device") = 0,
           "RegressionModel(X, y, start_at_mle): sufficient statistics are built on the "
           "device.  chains / seed / device: the many-chain engine behind the model.")
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
}
