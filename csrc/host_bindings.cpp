// pybind11 module `_host`: the parts of the runtime that need neither torch nor a GPU
// (rendezvous store, bucket planner).  Built with g++ in a few seconds, so the CPU test
// tier does not depend on the HIP toolchain.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "reducer/bucket_plan.h"
#include "store/tcp_store.h"

namespace py = pybind11;
using namespace dpa;

PYBIND11_MODULE(_host, m) {
  m.doc() = "host-side native runtime: TCP rendezvous store and gradient bucket planner";

  py::class_<TcpStoreServer>(m, "TcpStoreServer")
      .def(py::init<const std::string&, int>(), py::arg("bind_addr") = "0.0.0.0",
           py::arg("port") = 0)
      .def_property_readonly("port", &TcpStoreServer::port)
      .def("num_keys", &TcpStoreServer::num_keys)
      .def("shutdown", &TcpStoreServer::shutdown, py::call_guard<py::gil_scoped_release>());

  py::class_<TcpStoreClient>(m, "TcpStoreClient")
      .def(py::init<const std::string&, int, double>(), py::arg("host"), py::arg("port"),
           py::arg("timeout") = 300.0, py::call_guard<py::gil_scoped_release>())
      .def("set",
           [](TcpStoreClient& c, const std::string& k, py::bytes v) {
             std::string s = v;
             py::gil_scoped_release nogil;
             c.set(k, s);
           })
      .def(
          "get",
          [](TcpStoreClient& c, const std::string& k, double timeout) {
            std::string out;
            {
              py::gil_scoped_release nogil;
              out = c.get(k, timeout);
            }
            return py::bytes(out);
          },
          py::arg("key"), py::arg("timeout") = -1.0)
      .def("add", &TcpStoreClient::add, py::call_guard<py::gil_scoped_release>())
      .def("wait", &TcpStoreClient::wait, py::arg("keys"), py::arg("timeout") = -1.0,
           py::call_guard<py::gil_scoped_release>())
      .def("check", &TcpStoreClient::check, py::call_guard<py::gil_scoped_release>())
      .def("delete_key", &TcpStoreClient::delete_key, py::call_guard<py::gil_scoped_release>())
      .def("compare_set",
           [](TcpStoreClient& c, const std::string& k, py::bytes expected, py::bytes desired) {
             std::string e = expected, d = desired, out;
             {
               py::gil_scoped_release nogil;
               out = c.compare_set(k, e, d);
             }
             return py::bytes(out);
           })
      .def("num_keys", &TcpStoreClient::num_keys, py::call_guard<py::gil_scoped_release>())
      .def("append",
           [](TcpStoreClient& c, const std::string& k, py::bytes v) {
             std::string s = v;
             py::gil_scoped_release nogil;
             c.append(k, s);
           })
      .def("ping", &TcpStoreClient::ping, py::call_guard<py::gil_scoped_release>())
      .def("barrier", &TcpStoreClient::barrier, py::arg("name"), py::arg("world"),
           py::arg("timeout") = -1.0, py::call_guard<py::gil_scoped_release>())
      .def_property("timeout", &TcpStoreClient::timeout, &TcpStoreClient::set_timeout)
      .def_property_readonly("host", &TcpStoreClient::host)
      .def_property_readonly("port", &TcpStoreClient::port);

  py::class_<ParamMeta>(m, "ParamMeta")
      .def(py::init([](int64_t numel, int elem_size, int group_key, bool requires_grad) {
             ParamMeta p;
             p.numel = numel;
             p.elem_size = elem_size;
             p.group_key = group_key;
             p.requires_grad = requires_grad;
             return p;
           }),
           py::arg("numel"), py::arg("elem_size") = 4, py::arg("group_key") = 0,
           py::arg("requires_grad") = true)
      .def_readwrite("numel", &ParamMeta::numel)
      .def_readwrite("elem_size", &ParamMeta::elem_size)
      .def_readwrite("group_key", &ParamMeta::group_key)
      .def_readwrite("requires_grad", &ParamMeta::requires_grad);

  py::class_<Bucket>(m, "Bucket")
      .def_readonly("group_key", &Bucket::group_key)
      .def_readonly("elem_size", &Bucket::elem_size)
      .def_readonly("param_indices", &Bucket::param_indices)
      .def_readonly("offsets", &Bucket::offsets)
      .def_readonly("numel", &Bucket::numel)
      .def_readonly("arena_offset", &Bucket::arena_offset);

  py::class_<PlanOptions>(m, "PlanOptions")
      .def(py::init([](int64_t first_bucket_bytes, int64_t bucket_bytes, int64_t align_bytes,
                       bool reverse) {
             PlanOptions o;
             o.first_bucket_bytes = first_bucket_bytes;
             o.bucket_bytes = bucket_bytes;
             o.align_bytes = align_bytes;
             o.reverse = reverse;
             return o;
           }),
           py::arg("first_bucket_bytes") = (4ll << 20), py::arg("bucket_bytes") = (64ll << 20),
           py::arg("align_bytes") = 128, py::arg("reverse") = true)
      .def_readwrite("first_bucket_bytes", &PlanOptions::first_bucket_bytes)
      .def_readwrite("bucket_bytes", &PlanOptions::bucket_bytes)
      .def_readwrite("align_bytes", &PlanOptions::align_bytes)
      .def_readwrite("reverse", &PlanOptions::reverse);

  py::class_<Plan>(m, "Plan")
      .def_readonly("buckets", &Plan::buckets)
      .def_readonly("arena_numel", &Plan::arena_numel)
      .def("__repr__", [](const Plan& p) { return describe_plan(p); });

  m.def("plan_buckets", &plan_buckets, py::arg("params"), py::arg("options") = PlanOptions{});
  m.def("plan_buckets_from_order", &plan_buckets_from_order, py::arg("params"),
        py::arg("ready_order"), py::arg("options") = PlanOptions{});
}
