// cascade_amd/csrc/host/core.cpp
//
// Python bindings of cascade::sim (include/cascade/sim.hpp): module cascade_amd.core, the
// counterpart of the reference's cascade.core (cascade.py/core.cpp:49-359). Same class name,
// constructor keywords, properties, methods, outcome enum, conjunction dtype, zero-copy
// writable views of state / pars kept alive by shared_ptr capsules (:199-230), read-only
// conjunctions (:300-321), GIL released around construction, step() and propagate_until().
// `dyn` takes the descriptor returned by kepler() instead of heyoka expressions.
//
//   g++ -O2 -std=c++17 -fPIC -shared $(python3-config --includes) -I<pybind11> -Iinclude \
//       core.cpp -Lcascade_amd -lcascade -lcascade_hip -Wl,-rpath,'$ORIGIN' -o cascade_amd/core<ext>.so

#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <sstream>
#include <string>
#include <variant>
#include <vector>

#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cascade/logging.hpp>
#include <cascade/sim.hpp>

namespace py = pybind11;
using namespace cascade;
using namespace pybind11::literals;

namespace
{

struct coll_rec {
    std::uint64_t i, j;
    double time;
};

// A numpy view of a vector owned by a shared_ptr: the capsule keeps a copy of the pointer, so
// the array stays valid after the simulation replaces or drops the vector.
template <typename T>
py::array view_of(const std::shared_ptr<std::vector<T>> &sp, std::vector<py::ssize_t> shape, bool writable)
{
    auto *keep = new std::shared_ptr<std::vector<T>>(sp);
    py::capsule base(keep, [](void *p) { delete static_cast<std::shared_ptr<std::vector<T>> *>(p); });
    py::array ret(py::dtype::of<T>(), std::move(shape), sp->data(), base);
    if (!writable) {
        ret.attr("flags").attr("writeable") = false;
    }
    return ret;
}

std::vector<double> flatten2d(const py::array_t<double> &a)
{
    return py::cast<std::vector<double>>(a.attr("flatten")());
}

void check_state_array(const py::array_t<double> &state)
{
    if (state.ndim() != 2) {
        throw std::invalid_argument("The input state must have 2 dimensions, but instead an array with "
                                    + std::to_string(state.ndim()) + " dimension(s) was provided");
    }
    if (state.shape(1) != 7) {
        throw std::invalid_argument("An input state with 7 columns is expected, but the number of columns is instead "
                                    + std::to_string(state.shape(1)));
    }
}

std::vector<double> pars_vector(const std::optional<py::array_t<double>> &pars_, py::ssize_t nparts)
{
    std::vector<double> pars_vec;
    if (pars_) {
        const auto &pars = *pars_;
        if (pars.ndim() != 2) {
            throw std::invalid_argument("The input array of parameter values must have 2 dimensions, but "
                                        "instead an array with "
                                        + std::to_string(pars.ndim()) + " dimension(s) was provided");
        }
        if (pars.shape(0) != nparts) {
            throw std::invalid_argument("An input array of parameter values with " + std::to_string(nparts)
                                        + " row(s) is expected, but the number of rows is instead "
                                        + std::to_string(pars.shape(0)));
        }
        pars_vec = flatten2d(pars);
    }
    return pars_vec;
}

} // namespace

PYBIND11_MODULE(core, m)
{
    m.doc() = "MI355X-native cascade.core";

    // Expressions for user-defined dynamics (the role heyoka.py's expression class plays in the
    // reference's Python API, cascade.py/__init__.py:47-61): cascade_amd.dynamics re-exports them.
    py::class_<expression>(m, "expression")
        .def(py::init<double>())
        .def(py::init<std::string>())
        .def("__repr__",
             [](const expression &e) {
                 std::ostringstream oss;
                 oss << e;
                 return oss.str();
             })
        .def("__neg__", [](const expression &a) { return -a; })
        .def("__pos__", [](const expression &a) { return a; })
        .def("__add__", [](const expression &a, const expression &b) { return a + b; })
        .def("__add__", [](const expression &a, double b) { return a + expression(b); })
        .def("__radd__", [](const expression &a, double b) { return expression(b) + a; })
        .def("__sub__", [](const expression &a, const expression &b) { return a - b; })
        .def("__sub__", [](const expression &a, double b) { return a - expression(b); })
        .def("__rsub__", [](const expression &a, double b) { return expression(b) - a; })
        .def("__mul__", [](const expression &a, const expression &b) { return a * b; })
        .def("__mul__", [](const expression &a, double b) { return a * expression(b); })
        .def("__rmul__", [](const expression &a, double b) { return expression(b) * a; })
        .def("__truediv__", [](const expression &a, const expression &b) { return a / b; })
        .def("__truediv__", [](const expression &a, double b) { return a / expression(b); })
        .def("__rtruediv__", [](const expression &a, double b) { return expression(b) / a; })
        .def("__pow__", [](const expression &a, double e) { return pow(a, e); })
        .def("__pow__", [](const expression &a, const expression &e) { return pow(a, e); });
    py::implicitly_convertible<double, expression>();
    m.def("_ex_sin", [](const expression &a) { return sin(a); });
    m.def("_ex_cos", [](const expression &a) { return cos(a); });
    m.def("_ex_exp", [](const expression &a) { return exp(a); });
    m.def("_ex_sqrt", [](const expression &a) { return sqrt(a); });
    m.def("_ex_pow", [](const expression &a, double e) { return pow(a, e); });
    m.def("_ex_diff", [](const expression &e, const expression &v) { return diff(e, v); });
    m.def("_ex_par", [](std::uint32_t i) { return par(i); });
    m.def("_ex_time", []() { return time_expr; });

    // The lowered form of six equations: what the propagation kernel interprets (and what the
    // tests hand to the CPU oracle as the dynamics definition).
    PYBIND11_NUMPY_DTYPE(tape_op, op, dst, a, b, c);
    auto tape_dict = [](const std::vector<dynamics::dyn_t::eq_t> &eqs) {
        const tape_t t = lower_dynamics(eqs);
        py::array_t<tape_op> ops(static_cast<py::ssize_t>(t.ops.size()));
        std::copy(t.ops.begin(), t.ops.end(), ops.mutable_data());
        py::dict d;
        d["ops"] = ops;
        d["nslots"] = t.nslots;
        d["rhs"] = py::array_t<std::int32_t>(7, t.rhs.data());
        d["npars"] = t.npars;
        return d;
    };
    m.def("lower_dynamics", tape_dict);

    py::class_<dynamics::dyn_t>(m, "dyn_t")
        .def(py::init<std::vector<dynamics::dyn_t::eq_t>>())
        .def("__len__", &dynamics::dyn_t::size)
        .def("__getitem__",
             [](const dynamics::dyn_t &d, std::size_t i) {
                 if (i >= d.size()) {
                     throw py::index_error();
                 }
                 return d[i];
             })
        .def("__setitem__",
             [](dynamics::dyn_t &d, std::size_t i, dynamics::dyn_t::eq_t e) {
                 if (i >= d.size()) {
                     throw py::index_error();
                 }
                 d[i] = std::move(e);
             })
        .def_property_readonly("tape", [tape_dict](const dynamics::dyn_t &d) { return tape_dict(d.eqs); })
        .def_readonly("mu", &dynamics::dyn_t::mu)
        .def_readonly("npars", &dynamics::dyn_t::npars)
        .def_readonly("J2", &dynamics::dyn_t::J2)
        .def_readonly("Re", &dynamics::dyn_t::Re)
        .def_readonly("dens_alpha", &dynamics::dyn_t::dens_alpha)
        .def_readonly("dens_beta", &dynamics::dyn_t::dens_beta)
        .def_readonly("dens_gamma", &dynamics::dyn_t::dens_gamma)
        .def_property_readonly("kind", [](const dynamics::dyn_t &d) { return d.kind; })
        .def("__repr__", [](const dynamics::dyn_t &d) {
            std::ostringstream oss;
            if (d.kind == dynamics::dyn_t::tape) {
                oss << "dynamics(" << d.size() << " equations, interpreted from a tape)";
            } else {
                oss << "dynamics(kepler mu=" << d.mu << ((d.kind & 1) ? ", J2" : "") << ((d.kind & 2) ? ", drag" : "") << ")";
            }
            return oss.str();
        });
    m.def("_kepler", &dynamics::kepler, "mu"_a = 1.);
    m.def("_simple_earth", &dynamics::simple_earth, "J2"_a = true, "J3"_a = false, "J4"_a = false, "C22S22"_a = true,
          "sun"_a = false, "moon"_a = false, "SRP"_a = false, "drag"_a = true, "thermonets"_a = false);

    py::enum_<outcome>(m, "outcome")
        .value("success", outcome::success)
        .value("time_limit", outcome::time_limit)
        .value("collision", outcome::collision)
        .value("reentry", outcome::reentry)
        .value("exit", outcome::exit)
        .value("err_nf_state", outcome::err_nf_state);

    PYBIND11_NUMPY_DTYPE(sim::conjunction, i, j, time, dist, state_i, state_j);
    PYBIND11_NUMPY_DTYPE(coll_rec, i, j, time);

    using whitelist_t = sim::whitelist_t;
    py::class_<sim>(m, "sim", py::dynamic_attr{})
        .def(py::init<>())
        .def(py::init([](const py::array_t<double> &state, double ct,
                         std::optional<std::variant<dynamics::dyn_t, std::vector<dynamics::dyn_t::eq_t>>> dyn_,
                         std::optional<std::variant<double, std::vector<double>>> reentry_radius_,
                         std::optional<double> exit_radius_, const std::optional<py::array_t<double>> &pars_,
                         std::optional<double> tol_, bool ha, bool cm, std::uint32_t n_par_ct, double conj_thresh,
                         double min_coll_radius, whitelist_t coll_whitelist, whitelist_t conj_whitelist, int device) {
                 check_state_array(state);
                 sim::options o;
                 if (!dyn_) {
                     o.dyn = dynamics::kepler();
                 } else if (auto *dd = std::get_if<dynamics::dyn_t>(&*dyn_)) {
                     o.dyn = std::move(*dd);
                 } else {
                     o.dyn = dynamics::dyn_t(std::move(std::get<1>(*dyn_)));
                 }
                 o.reentry_radius = reentry_radius_ ? std::move(*reentry_radius_) : std::variant<double, std::vector<double>>{0.};
                 o.exit_radius = exit_radius_ ? *exit_radius_ : 0.;
                 o.pars = pars_vector(pars_, state.shape(0));
                 o.tol = tol_ ? *tol_ : 0.;
                 o.high_accuracy = ha;
                 o.compact_mode = cm;
                 o.n_par_ct = n_par_ct;
                 o.conj_thresh = conj_thresh;
                 o.min_coll_radius = min_coll_radius;
                 o.coll_whitelist = std::move(coll_whitelist);
                 o.conj_whitelist = std::move(conj_whitelist);
                 o.device = device;
                 auto state_vec = flatten2d(state);
                 py::gil_scoped_release release;
                 return sim(std::move(state_vec), ct, std::move(o));
             }),
             "state"_a, "ct"_a, "dyn"_a = py::none{}, "reentry_radius"_a = py::none{}, "exit_radius"_a = py::none{},
             "pars"_a = py::none{}, "tol"_a = py::none{}, "high_accuracy"_a = false, "compact_mode"_a = false,
             "n_par_ct"_a = 1, "conj_thresh"_a = 0., "min_coll_radius"_a = 0., "coll_whitelist"_a = whitelist_t{},
             "conj_whitelist"_a = whitelist_t{}, "device"_a = 0)
        .def_property_readonly("interrupt_info", &sim::get_interrupt_info)
        .def_property("time", &sim::get_time, &sim::set_time)
        .def_property("ct", &sim::get_ct, &sim::set_ct)
        .def_property("n_par_ct", &sim::get_n_par_ct, &sim::set_n_par_ct)
        .def_property("conj_thresh", &sim::get_conj_thresh, &sim::set_conj_thresh)
        .def_property("min_coll_radius", &sim::get_min_coll_radius, &sim::set_min_coll_radius)
        .def_property("coll_whitelist", &sim::get_coll_whitelist, &sim::set_coll_whitelist)
        .def_property("conj_whitelist", &sim::get_conj_whitelist, &sim::set_conj_whitelist)
        .def_property_readonly("nparts", &sim::get_nparts)
        .def_property_readonly("npars", &sim::get_npars)
        .def_property_readonly("tol", &sim::get_tol)
        .def_property_readonly("high_accuracy", &sim::get_high_accuracy)
        .def_property_readonly("compact_mode", &sim::get_compact_mode)
        .def_property_readonly("reentry_radius", &sim::get_reentry_radius)
        .def_property_readonly("exit_radius", &sim::get_exit_radius)
        .def(
            "step",
            [](sim &s) {
                py::gil_scoped_release release;
                return s.step();
            })
        .def(
            "propagate_until",
            [](sim &s, double t) {
                py::gil_scoped_release release;
                return s.propagate_until(t);
            },
            "t"_a)
        .def_property_readonly("state",
                               [](const sim &s) {
                                   auto sp = s._get_state_ptr();
                                   const auto n = static_cast<py::ssize_t>(sp->size() / 7u);
                                   return view_of(sp, {n, 7}, true);
                               })
        .def_property_readonly("pars",
                               [](const sim &s) {
                                   auto sp = s._get_pars_ptr();
                                   const auto np = static_cast<py::ssize_t>(s.get_npars());
                                   const auto n = static_cast<py::ssize_t>(s.get_nparts());
                                   return view_of(sp, {n, np}, true);
                               })
        .def_property_readonly("conjunctions",
                               [](const sim &s) {
                                   auto sp = s._get_conjunctions_ptr();
                                   return view_of(sp, {static_cast<py::ssize_t>(sp->size())}, false);
                               })
        .def("reset_conjunctions", &sim::reset_conjunctions)
        .def(
            "set_new_state_pars",
            [](sim &s, const py::array_t<double> &new_state, const std::optional<py::array_t<double>> &new_pars) {
                check_state_array(new_state);
                auto pv = pars_vector(new_pars, new_state.shape(0));
                auto sv = flatten2d(new_state);
                py::gil_scoped_release release;
                s.set_new_state_pars(std::move(sv), std::move(pv));
            },
            "new_state"_a, "new_pars"_a = py::none{})
        .def("remove_particles", &sim::remove_particles, "idxs"_a)
        .def("__repr__",
             [](const sim &s) {
                 std::ostringstream oss;
                 oss << s;
                 return oss.str();
             })
        // ---- additions of this build (measurement and residency control) ---------------
        .def("__copy__", [](const sim &s) { return sim(s); })
        .def("__deepcopy__", [](const sim &s, py::dict) { return sim(s); }, "memo"_a)
        .def_property_readonly("dyn", &sim::get_dyn)
        .def_property_readonly("device", &sim::get_device)
        .def_property("host_sync", &sim::get_host_sync, &sim::set_host_sync)
        .def_property("np_prefilter", &sim::get_np_prefilter, &sim::set_np_prefilter)
        .def("bp_nodes_visited", &sim::bp_nodes_visited)
        .def(
            "set_shard",
            [](sim &s, std::uint32_t rank, std::uint32_t nranks, py::object exchange) {
                // exchange(local) -> all: a numpy int64[6] in (the 48-byte summary), int64[nranks, 6] out.
                static_assert(sizeof(sim::shard_summary) == 6 * sizeof(std::int64_t), "summary layout");
                sim::exchange_t fn;
                if (!exchange.is_none()) {
                    fn = [exchange](const sim::shard_summary &mine, std::vector<sim::shard_summary> &all) {
                        py::gil_scoped_acquire gil;
                        py::array_t<std::int64_t> in(6);
                        std::memcpy(in.mutable_data(), &mine, sizeof(mine));
                        const auto out = exchange(in).cast<py::array_t<std::int64_t, py::array::c_style | py::array::forcecast>>();
                        if (out.size() % 6 != 0) {
                            throw std::runtime_error("sharded simulation: malformed exchange result");
                        }
                        all.resize(static_cast<std::size_t>(out.size() / 6));
                        std::memcpy(static_cast<void *>(all.data()), out.data(), all.size() * sizeof(sim::shard_summary));
                    };
                }
                s.set_shard(rank, nranks, std::move(fn));
            },
            "rank"_a, "nranks"_a, "exchange"_a = py::none{})
        .def(
            "set_shard_propagation",
            [](sim &s, py::object gather, bool chunks, py::object alltoall) {
                // gather(send_address, recv_address, nbytes): all-gather of device memory;
                // alltoall(send_address, recv_address, send_bytes list, recv_bytes_each).
                sim::gather_t fn;
                if (!gather.is_none()) {
                    fn = [gather](std::uintptr_t send, std::uintptr_t recv, std::size_t bytes) {
                        py::gil_scoped_acquire gil;
                        gather(send, recv, bytes);
                    };
                }
                sim::alltoall_t fa;
                if (!alltoall.is_none()) {
                    fa = [alltoall](std::uintptr_t send, std::uintptr_t recv, const std::vector<std::size_t> &seg, std::size_t each) {
                        py::gil_scoped_acquire gil;
                        alltoall(send, recv, seg, each);
                    };
                }
                s.set_shard_propagation(std::move(fn), chunks, std::move(fa));
            },
            "gather"_a = py::none{}, "chunks"_a = true, "alltoall"_a = py::none{})
        .def_property_readonly("last_exchanged_bytes", [](const sim &s) { return s.get_step_stats().exchanged_bytes; })
        .def_property_readonly("last_chunk_block", [](const sim &s) { return std::make_pair(s.get_step_stats().chunk_first, s.get_step_stats().chunk_count); })
        .def_property_readonly("_conj_segments",
                               [](const sim &s) {
                                   const auto &v = s._get_conj_segments();
                                   py::array_t<std::uint64_t> out({static_cast<py::ssize_t>(v.size()), static_cast<py::ssize_t>(2)});
                                   auto *p = out.mutable_data();
                                   for (std::size_t k = 0; k < v.size(); ++k) {
                                       p[2 * k] = v[k].first;
                                       p[2 * k + 1] = v[k].second;
                                   }
                                   return out;
                               })
        .def("upload_state", &sim::upload_state)
        .def("download_state", &sim::download_state)
        .def_property_readonly("last_bp_pairs", [](const sim &s) { return s.get_step_stats().bp_pairs; })
        .def_property_readonly("last_np_candidates", [](const sim &s) { return s.get_step_stats().np_candidates; })
        .def_property_readonly("last_stored_objects", [](const sim &s) { return s.get_step_stats().stored_objects; })
        .def_property_readonly("last_np_attempts", [](const sim &s) { return s.get_step_stats().np_attempts; })
        .def_property_readonly("last_propagated", [](const sim &s) { return s.get_step_stats().propagated; })
        .def_property_readonly("last_bp_node_pairs", [](const sim &s) { return s.get_step_stats().bp_node_pairs; })
        .def_property_readonly("last_bp_tiles", [](const sim &s) { return s.get_step_stats().bp_tiles; })
        .def_property_readonly("last_total_substeps", [](const sim &s) { return s.get_step_stats().total_substeps; })
        .def_property_readonly("last_nchunks", [](const sim &s) { return s.get_step_stats().nchunks; })
        .def_property_readonly("last_collisions",
                               [](const sim &s) {
                                   const auto &v = s.get_last_collisions();
                                   py::array_t<coll_rec> out(static_cast<py::ssize_t>(v.size()));
                                   auto *p = static_cast<coll_rec *>(out.mutable_data());
                                   for (std::size_t k = 0; k < v.size(); ++k) {
                                       p[k] = coll_rec{std::get<0>(v[k]), std::get<1>(v[k]), std::get<2>(v[k])};
                                   }
                                   return out;
                               })
        .def_property_readonly("last_step_conjunctions",
                               [](const sim &s) {
                                   const auto &v = s.get_last_step_conjunctions();
                                   py::array_t<sim::conjunction> out(static_cast<py::ssize_t>(v.size()));
                                   std::copy(v.begin(), v.end(), static_cast<sim::conjunction *>(out.mutable_data()));
                                   return out;
                               })
        .def_property_readonly("_hip_context", [](sim &s) { return reinterpret_cast<std::uintptr_t>(s._get_hip_context()); });

    // Thread-count control of the reference (cascade.py/core.cpp:332-348): the pipeline runs on
    // the GPU, so these only record the value.
    static unsigned nthreads = 0;
    m.def("set_nthreads", [](unsigned n) { nthreads = n; }, "n"_a);
    m.def("get_nthreads", []() { return nthreads; });
    // Logger levels of the reference (cascade.py/logging.cpp:19-29).
    m.def("set_logger_level_trace", &cascade::set_logger_level_trace);
    m.def("set_logger_level_debug", &cascade::set_logger_level_debug);
    m.def("set_logger_level_info", &cascade::set_logger_level_info);
    m.def("set_logger_level_warn", &cascade::set_logger_level_warn);
    m.def("set_logger_level_err", &cascade::set_logger_level_err);
    m.def("set_logger_level_critical", &cascade::set_logger_level_critical);
}
