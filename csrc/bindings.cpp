// pybind11 module `_C`: RCCL communicator, gradient reducer and (via ops_bindings.cpp) the
// HIP kernels, all operating on torch tensors and torch's current HIP stream.
#include <torch/extension.h>

#include "comm/rccl_comm.h"
#include "reducer/reducer.h"
#include "trace.h"

namespace py = pybind11;
using namespace dpa;

void init_ops(py::module_& m);  // ops_bindings.cpp

PYBIND11_MODULE(_C, m) {
  m.doc() = "gfx950 native runtime: RCCL communicator, gradient reducer, HIP kernels";
  m.attr("gpu_arch") = "gfx950";

  py::class_<RcclComm, std::shared_ptr<RcclComm>>(m, "RcclComm")
      .def_static("unique_id", []() { return py::bytes(RcclComm::unique_id()); })
      .def_static("device_bus_id", &RcclComm::device_bus_id, py::arg("device"))
      .def_static("version", &RcclComm::version)
      .def(py::init([](py::bytes uid, int rank, int world, int device, bool high_priority_stream) {
             std::string s = uid;
             py::gil_scoped_release nogil;  // blocks until every rank has joined
             return std::make_shared<RcclComm>(s, rank, world, device, high_priority_stream);
           }),
           py::arg("uid"), py::arg("rank"), py::arg("world"), py::arg("device"), py::arg("high_priority_stream") = false)
      .def_property_readonly("rank", &RcclComm::rank)
      .def_property_readonly("world", &RcclComm::world)
      .def_property_readonly("device", &RcclComm::device)
      .def("all_reduce", &RcclComm::all_reduce, py::arg("tensor"), py::arg("op") = 0)
      .def("reduce", &RcclComm::reduce, py::arg("tensor"), py::arg("root") = 0, py::arg("op") = 0)
      .def("broadcast", &RcclComm::broadcast, py::arg("tensor"), py::arg("root") = 0)
      .def("gather", &RcclComm::gather, py::arg("input"), py::arg("output"), py::arg("root") = 0)
      .def("all_gather", &RcclComm::all_gather)
      .def("reduce_scatter", &RcclComm::reduce_scatter, py::arg("input"), py::arg("output"), py::arg("op") = 0)
      .def("barrier", &RcclComm::barrier, py::call_guard<py::gil_scoped_release>())
      .def("all_reduce_async", &RcclComm::all_reduce_async, py::arg("tensor"), py::arg("op"), py::arg("slot"))
      .def("broadcast_async", &RcclComm::broadcast_async, py::arg("tensor"), py::arg("root"), py::arg("slot"))
      .def("reduce_scatter_async", &RcclComm::reduce_scatter_async, py::arg("tensor"), py::arg("op"), py::arg("slot"))
      .def("all_gather_async", &RcclComm::all_gather_async, py::arg("tensor"), py::arg("slot"))
      .def("wait_slot", &RcclComm::wait_slot)
      .def("wait_all_slots", &RcclComm::wait_all_slots)
      .def("synchronize_comm_stream", &RcclComm::synchronize_comm_stream, py::call_guard<py::gil_scoped_release>())
      .def("comm_stream_handle", [](RcclComm& c) { return reinterpret_cast<uintptr_t>(c.comm_stream()); })
      .def("set_comm_stream", &RcclComm::set_comm_stream, py::arg("handle"))
      .def("comm_stream_info",
           [](RcclComm& c) {
             c.comm_stream();
             const CommStreamChoice& ch = c.comm_stream_choice();
             py::dict d;
             d["priority"] = ch.high_priority ? "high" : "normal";
             d["independent"] = ch.independent;
             d["candidates_tried"] = ch.candidates_tried;
             d["reused"] = ch.reused;
             d["probe_ms"] = ch.probe_ms;
             d["text"] = ch.describe();
             return d;
           })
      .def("aborted", &RcclComm::aborted)
      .def("set_emulation", &RcclComm::set_emulation, py::arg("ranks"), py::arg("busbw_gbps"), py::arg("workgroups") = 32,
           py::arg("passes") = 2, py::arg("latency_us") = 30.0)
      .def("emulating", &RcclComm::emulating)
      .def("set_one_rank_identity", &RcclComm::set_one_rank_identity, py::arg("on"))
      .def("one_rank_identity", &RcclComm::one_rank_identity)
      .def("group_start", &RcclComm::group_start)
      .def("group_end", &RcclComm::group_end)
      .def("num_async_ops", &RcclComm::num_async_ops)
      .def("destroy", &RcclComm::destroy, py::call_guard<py::gil_scoped_release>())
      .def("abort", &RcclComm::abort, py::call_guard<py::gil_scoped_release>())
      .def("set_debug_sync", &RcclComm::set_debug_sync, py::arg("on"))
      .def("debug_sync", &RcclComm::debug_sync)
      .def("comm_stream_idle", &RcclComm::comm_stream_idle)
      .def("set_timing", &RcclComm::set_timing, py::arg("on"))
      .def("read_timing", &RcclComm::read_timing, py::call_guard<py::gil_scoped_release>());

  m.def(
      "probe_stream_independence",
      [](int device, uintptr_t candidate, uintptr_t compute, double hold_ms) {
        py::gil_scoped_release nogil;
        return probe_stream_independence(device, reinterpret_cast<hipStream_t>(candidate), reinterpret_cast<hipStream_t>(compute),
                                         hold_ms);
      },
      py::arg("device"), py::arg("candidate"), py::arg("compute"), py::arg("hold_ms") = 3.0,
      "Milliseconds a one-lane kernel on `compute` needed while a hold_ms spin ran on `candidate`");

  py::class_<Transport, std::shared_ptr<Transport>>(m, "Transport")
      .def("native_average", &Transport::native_average)
      .def("rank", &Transport::rank)
      .def("world", &Transport::world);
  m.def("make_rccl_transport", &make_rccl_transport);
  m.def("make_c10d_transport", &make_c10d_transport);
  m.def("make_local_transport", &make_local_transport);

  py::class_<BucketSpec>(m, "BucketSpec")
      .def(py::init([](std::vector<int64_t> idx, std::vector<int64_t> offs, at::Tensor buffer) {
             BucketSpec s;
             s.param_indices = std::move(idx);
             s.offsets = std::move(offs);
             s.buffer = std::move(buffer);
             return s;
           }),
           py::arg("param_indices"), py::arg("offsets"), py::arg("buffer"))
      .def_readonly("param_indices", &BucketSpec::param_indices)
      .def_readonly("offsets", &BucketSpec::offsets)
      .def_readonly("buffer", &BucketSpec::buffer);

  py::class_<Reducer, std::shared_ptr<Reducer>>(m, "Reducer")
      .def(py::init([](std::vector<at::Tensor> params, std::vector<BucketSpec> buckets, std::shared_ptr<Transport> t,
                       bool gradient_as_bucket_view, bool unpack_to_grads, int64_t max_collective_bytes, bool track_unused,
                       bool shard_gradients) {
             ReducerOptions o;
             o.shard_gradients = shard_gradients;
             o.gradient_as_bucket_view = gradient_as_bucket_view;
             o.unpack_to_grads = unpack_to_grads;
             o.max_collective_bytes = max_collective_bytes;
             o.track_unused = track_unused;
             return std::make_shared<Reducer>(std::move(params), std::move(buckets), std::move(t), o);
           }),
           py::arg("params"), py::arg("buckets"), py::arg("transport"), py::arg("gradient_as_bucket_view") = true,
           py::arg("unpack_to_grads") = true, py::arg("max_collective_bytes") = 0,
           py::arg("track_unused") = false, py::arg("shard_gradients") = false)
      // Everything that can wait for a collective releases the GIL: the host-side (gloo) group finishes its work
      // objects on worker threads that take the GIL to drop tensors handed in from Python; a caller that blocks
      // with the GIL held can leave both of those threads waiting for it and nobody to run the collective.
      .def("prepare_for_backward", &Reducer::prepare_for_backward, py::arg("sync") = true, py::call_guard<py::gil_scoped_release>())
      .def("armed", &Reducer::armed)
      .def("ready_order", &Reducer::ready_order)
      .def("num_buckets", &Reducer::num_buckets)
      .def("num_backward_syncs", &Reducer::num_backward_syncs)
      .def("num_bucket_launches", &Reducer::num_bucket_launches)
      .def("num_pack_launches", &Reducer::num_pack_launches)
      .def("num_collectives", &Reducer::num_collectives)
      .def("launch_positions", &Reducer::launch_positions)
      .def("slices_per_bucket", &Reducer::slices_per_bucket)
      .def("globally_unused", &Reducer::globally_unused)
      .def("take_grad_sink",
           [](Reducer& r, int64_t i) -> py::object {
             at::Tensor t = r.take_grad_sink(i);
             return t.defined() ? py::cast(t) : py::none();
           })
      .def("set_deferred_wait", &Reducer::set_deferred_wait, py::arg("on"))
      .def("deferred_wait", &Reducer::deferred_wait)
      .def("waits_pending", &Reducer::waits_pending)
      .def("wait_bucket", &Reducer::wait_bucket, py::arg("bucket"), py::call_guard<py::gil_scoped_release>())
      .def("wait_pending", &Reducer::wait_pending, py::call_guard<py::gil_scoped_release>())
      .def("bucket_of", &Reducer::bucket_of, py::arg("param_index"))
      .def("all_gather_bucket", &Reducer::all_gather_bucket, py::arg("bucket"), py::arg("flat"), py::call_guard<py::gil_scoped_release>())
      .def("num_grads_in_place", &Reducer::num_grads_in_place)
      .def("num_grads_packed", &Reducer::num_grads_packed)
      .def("broadcast_parameters", &Reducer::broadcast_parameters, py::arg("root") = 0,
           py::call_guard<py::gil_scoped_release>())
      .def("detach", &Reducer::detach);

  // Profiler ranges (roctx), shared with the reducer's own ranges.
  m.def("trace_enable", &trace::enable, py::arg("on"));
  m.def("trace_enabled", &trace::enabled);
  m.def("trace_push", [](const std::string& name) { trace::push(name); });
  m.def("trace_pop", []() { trace::pop(); });

  init_ops(m);
}
