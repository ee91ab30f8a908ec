#include "rccl_comm.h"

#include "kernels/kernels.h"

#include <c10/hip/HIPStream.h>

#include <chrono>
#include <cstring>
#include <sstream>
#include <stdexcept>

namespace dpa {

namespace {

#define DPA_HIP_CHECK(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      std::ostringstream _os;                                                                \
      _os << "HIP error " << hipGetErrorString(_e) << " at " << __FILE__ << ":" << __LINE__  \
          << " in " #expr;                                                                   \
      throw std::runtime_error(_os.str());                                                   \
    }                                                                                        \
  } while (0)

#define DPA_NCCL_CHECK(expr)                                                                 \
  do {                                                                                       \
    ncclResult_t _r = (expr);                                                                \
    if (_r != ncclSuccess && _r != ncclInProgress) {                                         \
      std::ostringstream _os;                                                                \
      _os << "RCCL error " << ncclGetErrorString(_r) << " at " << __FILE__ << ":" << __LINE__ \
          << " in " #expr;                                                                   \
      throw std::runtime_error(_os.str());                                                   \
    }                                                                                        \
  } while (0)

hipStream_t current_stream(int device) {
  return c10::hip::getCurrentHIPStream(static_cast<c10::DeviceIndex>(device)).stream();
}

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

ncclDataType_t to_nccl_dtype(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return ncclFloat32;
    case at::kDouble: return ncclFloat64;
    case at::kHalf: return ncclFloat16;
    case at::kBFloat16: return ncclBfloat16;
    case at::kInt: return ncclInt32;
    case at::kLong: return ncclInt64;
    case at::kByte: return ncclUint8;
    case at::kChar: return ncclInt8;
    case at::kBool: return ncclUint8;
    default: throw std::runtime_error(std::string("RcclComm: unsupported dtype ") + c10::toString(t));
  }
}

std::string RcclComm::unique_id() {
  ncclUniqueId id;
  DPA_NCCL_CHECK(ncclGetUniqueId(&id));
  return std::string(reinterpret_cast<const char*>(&id), sizeof(id));
}

std::string RcclComm::device_bus_id(int device) {
  char buf[64] = {0};
  DPA_HIP_CHECK(hipDeviceGetPCIBusId(buf, sizeof(buf), device));
  return std::string(buf);
}

std::string RcclComm::version() {
  int v = 0;
  DPA_NCCL_CHECK(ncclGetVersion(&v));
  std::ostringstream os;
  os << v / 10000 << "." << (v / 100) % 100 << "." << v % 100;
  return os.str();
}

RcclComm::RcclComm(const std::string& uid, int rank, int world, int device, bool high_priority_stream)
    : rank_(rank), world_(world), device_(device), want_high_(high_priority_stream) {
  if (uid.size() != sizeof(ncclUniqueId)) throw std::runtime_error("RcclComm: bad unique id size");
  ncclUniqueId id;
  std::memcpy(&id, uid.data(), sizeof(id));
  DPA_HIP_CHECK(hipSetDevice(device_));
  DPA_NCCL_CHECK(ncclCommInitRank(&comm_, world_, id, rank_));
  // The communication stream is the process's one long-lived stream for this device (comm_stream.h): chosen
  // by a bounded probe at the first communicator, reused by every later one.
  ensure_comm_stream();
  barrier_scratch_ = at::zeros({1}, at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, device_));
}

RcclComm::~RcclComm() {
  try {
    destroy();
  } catch (...) {
  }
}

void RcclComm::destroy() {
  if (destroyed_) return;
  destroyed_ = true;
  DeviceGuard g(device_);
  if (comm_stream_ && !aborted_.load()) (void)hipStreamSynchronize(comm_stream_);
  {
    std::lock_guard<std::timed_mutex> lk(comm_mu_);
    if (comm_) {
      (void)ncclCommFinalize(comm_);
      (void)ncclCommDestroy(comm_);
      comm_ = nullptr;
    }
  }
  for (hipEvent_t e : slot_events_)
    if (e) (void)hipEventDestroy(e);
  slot_events_.clear();
  for (hipEvent_t e : fence_events_)
    if (e) (void)hipEventDestroy(e);
  fence_events_.clear();
  for (hipEvent_t e : timing_pool_)
    if (e) (void)hipEventDestroy(e);
  timing_pool_.clear();
  op_spans_.clear();
  wait_spans_.clear();
  comm_stream_ = nullptr;  // the stream itself belongs to the process (comm_stream.h) and stays
  barrier_scratch_ = at::Tensor();
}

void RcclComm::abort() {
  aborted_.store(true);  // from here on no enqueue path touches comm_ any more
  std::unique_lock<std::timed_mutex> lk(comm_mu_, std::chrono::milliseconds(2000));
  // Not acquired: the training thread is blocked inside a library call that holds the mutex -- the very
  // situation abort is for; ncclCommAbort is what releases it.
  ncclComm_t c = comm_;
  comm_ = nullptr;
  if (c) (void)ncclCommAbort(c);
}

void RcclComm::check_tensor(const at::Tensor& t) const {
  if (aborted_.load()) throw std::runtime_error("RcclComm: communicator was aborted");
  if (destroyed_ || !comm_) throw std::runtime_error("RcclComm: communicator already destroyed");
  if (!t.is_cuda()) throw std::runtime_error("RcclComm: tensor must live on the GPU");
  if (t.get_device() != device_) {
    std::ostringstream os;
    os << "RcclComm: tensor is on cuda:" << t.get_device() << " but this rank's communicator is bound to cuda:" << device_
       << ".  With the launcher's default GPU binding (DIST_GPU_BINDING=rotate) every rank sees ITS GPU as cuda:0 -- use "
          "distributed.get_device() instead of f\"cuda:{rank}\", or launch with DIST_GPU_BINDING=index to keep device "
          "index == rank as in the reference";
    throw std::runtime_error(os.str());
  }
  if (!t.is_contiguous()) throw std::runtime_error("RcclComm: tensor must be contiguous");
}

void RcclComm::all_reduce(at::Tensor t, int op) {
  check_tensor(t);
  if (t.numel() == 0) return;
  DeviceGuard g(device_);
  std::lock_guard<std::timed_mutex> lk(comm_mu_);
  if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
  DPA_NCCL_CHECK(ncclAllReduce(t.data_ptr(), t.data_ptr(), static_cast<size_t>(t.numel()), to_nccl_dtype(t.scalar_type()),
                               static_cast<ncclRedOp_t>(op), comm_, current_stream(device_)));
}

void RcclComm::reduce(at::Tensor t, int root, int op) {
  check_tensor(t);
  if (t.numel() == 0) return;
  DeviceGuard g(device_);
  std::lock_guard<std::timed_mutex> lk(comm_mu_);
  if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
  DPA_NCCL_CHECK(ncclReduce(t.data_ptr(), t.data_ptr(), static_cast<size_t>(t.numel()), to_nccl_dtype(t.scalar_type()),
                            static_cast<ncclRedOp_t>(op), root, comm_, current_stream(device_)));
}

void RcclComm::broadcast(at::Tensor t, int root) {
  check_tensor(t);
  if (t.numel() == 0) return;
  DeviceGuard g(device_);
  std::lock_guard<std::timed_mutex> lk(comm_mu_);
  if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
  DPA_NCCL_CHECK(ncclBroadcast(t.data_ptr(), t.data_ptr(), static_cast<size_t>(t.numel()), to_nccl_dtype(t.scalar_type()),
                               root, comm_, current_stream(device_)));
}

void RcclComm::gather(at::Tensor in, c10::optional<at::Tensor> out, int root) {
  check_tensor(in);
  void* recv = nullptr;
  if (rank_ == root) {
    if (!out.has_value()) throw std::runtime_error("RcclComm::gather: root needs an output tensor");
    check_tensor(*out);
    if (out->numel() != in.numel() * world_) throw std::runtime_error("RcclComm::gather: output must hold world * input elements");
    recv = out->data_ptr();
  }
  if (in.numel() == 0) return;
  DeviceGuard g(device_);
  std::lock_guard<std::timed_mutex> lk(comm_mu_);
  if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
  DPA_NCCL_CHECK(ncclGather(in.data_ptr(), recv, static_cast<size_t>(in.numel()), to_nccl_dtype(in.scalar_type()), root,
                            comm_, current_stream(device_)));
}

void RcclComm::all_gather(at::Tensor in, at::Tensor out) {
  check_tensor(in);
  check_tensor(out);
  if (out.numel() != in.numel() * world_) throw std::runtime_error("RcclComm::all_gather: output must hold world * input elements");
  if (in.numel() == 0) return;
  DeviceGuard g(device_);
  std::lock_guard<std::timed_mutex> lk(comm_mu_);
  if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
  DPA_NCCL_CHECK(ncclAllGather(in.data_ptr(), out.data_ptr(), static_cast<size_t>(in.numel()), to_nccl_dtype(in.scalar_type()),
                               comm_, current_stream(device_)));
}

void RcclComm::reduce_scatter(at::Tensor in, at::Tensor out, int op) {
  check_tensor(in);
  check_tensor(out);
  if (in.numel() != out.numel() * world_) throw std::runtime_error("RcclComm::reduce_scatter: input must hold world * output elements");
  if (out.numel() == 0) return;
  DeviceGuard g(device_);
  std::lock_guard<std::timed_mutex> lk(comm_mu_);
  if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
  DPA_NCCL_CHECK(ncclReduceScatter(in.data_ptr(), out.data_ptr(), static_cast<size_t>(out.numel()),
                                   to_nccl_dtype(in.scalar_type()), static_cast<ncclRedOp_t>(op), comm_,
                                   current_stream(device_)));
}

void RcclComm::barrier() {
  if (destroyed_ || !comm_) throw std::runtime_error("RcclComm: communicator already destroyed");
  DeviceGuard g(device_);
  hipStream_t s = current_stream(device_);
  {
    std::lock_guard<std::timed_mutex> lk(comm_mu_);
    if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
    DPA_NCCL_CHECK(ncclAllReduce(barrier_scratch_.data_ptr(), barrier_scratch_.data_ptr(), 1, ncclFloat32, ncclSum, comm_, s));
  }
  DPA_HIP_CHECK(hipStreamSynchronize(s));  // outside the mutex: abort() must be able to get in while this blocks
}

// Prepares slot `slot` for a new operation: events exist, and a previous operation on the same
// slot that nobody waited for has finished.  An event is never re-recorded while a stream may
// still be waiting on its previous record: every slot owns its own fence and completion event.
void RcclComm::ensure_comm_stream() {
  if (!comm_stream_) {
    choice_ = acquire_comm_stream(device_, current_stream(device_), want_high_);
    comm_stream_ = choice_.stream;
  }
}

void RcclComm::set_comm_stream(uintptr_t handle) {
  if (comm_stream_) (void)hipStreamSynchronize(comm_stream_);
  comm_stream_ = reinterpret_cast<hipStream_t>(handle);
  choice_ = CommStreamChoice();
  choice_.stream = comm_stream_;
}

void RcclComm::begin_slot(int slot) {
  if (slot < 0) throw std::runtime_error("RcclComm: negative slot");
  ensure_comm_stream();
  const size_t s = static_cast<size_t>(slot);
  if (s >= slot_events_.size()) {
    slot_events_.resize(s + 1, nullptr);
    fence_events_.resize(s + 1, nullptr);
    slot_pending_.resize(s + 1, 0);
  }
  if (!slot_events_[s]) DPA_HIP_CHECK(hipEventCreateWithFlags(&slot_events_[s], hipEventDisableTiming));
  if (!fence_events_[s]) DPA_HIP_CHECK(hipEventCreateWithFlags(&fence_events_[s], hipEventDisableTiming));
  if (slot_pending_[s]) DPA_HIP_CHECK(hipEventSynchronize(slot_events_[s]));
}

hipEvent_t RcclComm::timing_event() {
  if (timing_used_ >= 16384) {
    // Nobody read the window: start a new one instead of growing without bound.
    op_spans_.clear();
    wait_spans_.clear();
    timing_used_ = 0;
  }
  if (timing_used_ == timing_pool_.size()) {
    hipEvent_t e;
    DPA_HIP_CHECK(hipEventCreate(&e));
    timing_pool_.push_back(e);
  }
  return timing_pool_[timing_used_++];
}

void RcclComm::finish_async(int slot, hipEvent_t done, hipEvent_t t0) {
  if (t0) {
    hipEvent_t t1 = timing_event();
    DPA_HIP_CHECK(hipEventRecord(t1, comm_stream_));
    op_spans_.emplace_back(t0, t1);
  }
  DPA_HIP_CHECK(hipEventRecord(done, comm_stream_));
  slot_pending_[static_cast<size_t>(slot)] = 1;
  ++n_async_;
  if (debug_sync_) DPA_HIP_CHECK(hipStreamSynchronize(comm_stream_));
}

void RcclComm::all_reduce_async(at::Tensor t, int op, int slot) {
  check_tensor(t);
  DeviceGuard g(device_);
  begin_slot(slot);
  hipEvent_t done = slot_events_[static_cast<size_t>(slot)], fence = fence_events_[static_cast<size_t>(slot)];
  // Fence: the bucket was filled by kernels queued on the current stream.
  DPA_HIP_CHECK(hipEventRecord(fence, current_stream(device_)));
  DPA_HIP_CHECK(hipStreamWaitEvent(comm_stream_, fence, 0));
  hipEvent_t t0 = nullptr;
  if (timing_) {
    t0 = timing_event();
    DPA_HIP_CHECK(hipEventRecord(t0, comm_stream_));
  }
  if (t.numel() > 0 && emulating()) {
    const double bytes = static_cast<double>(t.numel()) * static_cast<double>(t.element_size());
    const double ms = 2.0 * (emu_ranks_ - 1) / emu_ranks_ * bytes / (emu_busbw_gbps_ * 1e9) * 1e3 + emu_latency_us_ * 1e-3;
    kernels::launch_emulated_collective(t.data_ptr(), static_cast<long long>(bytes), emu_workgroups_, emu_passes_, ms, comm_stream_);
  } else if (t.numel() > 0 && !one_rank_identity_) {
    std::lock_guard<std::timed_mutex> lk(comm_mu_);
    if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
    DPA_NCCL_CHECK(ncclAllReduce(t.data_ptr(), t.data_ptr(), static_cast<size_t>(t.numel()), to_nccl_dtype(t.scalar_type()),
                                 static_cast<ncclRedOp_t>(op), comm_, comm_stream_));
  }
  finish_async(slot, done, t0);
}

void RcclComm::broadcast_async(at::Tensor t, int root, int slot) {
  check_tensor(t);
  DeviceGuard g(device_);
  begin_slot(slot);
  hipEvent_t done = slot_events_[static_cast<size_t>(slot)], fence = fence_events_[static_cast<size_t>(slot)];
  DPA_HIP_CHECK(hipEventRecord(fence, current_stream(device_)));
  DPA_HIP_CHECK(hipStreamWaitEvent(comm_stream_, fence, 0));
  hipEvent_t t0 = nullptr;
  if (timing_) {
    t0 = timing_event();
    DPA_HIP_CHECK(hipEventRecord(t0, comm_stream_));
  }
  if (t.numel() > 0) {
    std::lock_guard<std::timed_mutex> lk(comm_mu_);
    if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
    DPA_NCCL_CHECK(ncclBroadcast(t.data_ptr(), t.data_ptr(), static_cast<size_t>(t.numel()), to_nccl_dtype(t.scalar_type()),
                                 root, comm_, comm_stream_));
  }
  finish_async(slot, done, t0);
}

void RcclComm::reduce_scatter_async(at::Tensor t, int op, int slot) {
  check_tensor(t);
  if (t.numel() % world_ != 0) throw std::runtime_error("RcclComm::reduce_scatter_async: buffer must hold world * shard elements");
  DeviceGuard g(device_);
  begin_slot(slot);
  hipEvent_t done = slot_events_[static_cast<size_t>(slot)], fence = fence_events_[static_cast<size_t>(slot)];
  DPA_HIP_CHECK(hipEventRecord(fence, current_stream(device_)));
  DPA_HIP_CHECK(hipStreamWaitEvent(comm_stream_, fence, 0));
  hipEvent_t t0 = nullptr;
  if (timing_) {
    t0 = timing_event();
    DPA_HIP_CHECK(hipEventRecord(t0, comm_stream_));
  }
  const size_t shard = static_cast<size_t>(t.numel() / world_);
  if (shard > 0 && emulating()) {
    const double bytes = static_cast<double>(t.numel()) * static_cast<double>(t.element_size());
    const double ms = (emu_ranks_ - 1.0) / emu_ranks_ * bytes / (emu_busbw_gbps_ * 1e9) * 1e3 + emu_latency_us_ * 1e-3;
    kernels::launch_emulated_collective(t.data_ptr(), static_cast<long long>(bytes), emu_workgroups_, 1, ms, comm_stream_);
  } else if (shard > 0 && !one_rank_identity_) {
    std::lock_guard<std::timed_mutex> lk(comm_mu_);
    if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
    char* base = static_cast<char*>(t.data_ptr());
    DPA_NCCL_CHECK(ncclReduceScatter(base, base + static_cast<size_t>(rank_) * shard * t.element_size(), shard, to_nccl_dtype(t.scalar_type()),
                                     static_cast<ncclRedOp_t>(op), comm_, comm_stream_));
  }
  finish_async(slot, done, t0);
}

void RcclComm::all_gather_async(at::Tensor t, int slot) {
  check_tensor(t);
  if (t.numel() % world_ != 0) throw std::runtime_error("RcclComm::all_gather_async: buffer must hold world * shard elements");
  DeviceGuard g(device_);
  begin_slot(slot);
  hipEvent_t done = slot_events_[static_cast<size_t>(slot)], fence = fence_events_[static_cast<size_t>(slot)];
  DPA_HIP_CHECK(hipEventRecord(fence, current_stream(device_)));
  DPA_HIP_CHECK(hipStreamWaitEvent(comm_stream_, fence, 0));
  hipEvent_t t0 = nullptr;
  if (timing_) {
    t0 = timing_event();
    DPA_HIP_CHECK(hipEventRecord(t0, comm_stream_));
  }
  const size_t shard = static_cast<size_t>(t.numel() / world_);
  if (shard > 0 && emulating()) {
    const double bytes = static_cast<double>(t.numel()) * static_cast<double>(t.element_size());
    const double ms = (emu_ranks_ - 1.0) / emu_ranks_ * bytes / (emu_busbw_gbps_ * 1e9) * 1e3 + emu_latency_us_ * 1e-3;
    kernels::launch_emulated_collective(t.data_ptr(), static_cast<long long>(bytes), emu_workgroups_, 1, ms, comm_stream_);
  } else if (shard > 0 && !one_rank_identity_) {
    std::lock_guard<std::timed_mutex> lk(comm_mu_);
    if (!comm_) throw std::runtime_error("RcclComm: communicator was aborted");
    char* base = static_cast<char*>(t.data_ptr());
    DPA_NCCL_CHECK(ncclAllGather(base + static_cast<size_t>(rank_) * shard * t.element_size(), base, shard, to_nccl_dtype(t.scalar_type()),
                                 comm_, comm_stream_));
  }
  finish_async(slot, done, t0);
}

void RcclComm::wait_slot(int slot) {
  if (slot < 0 || static_cast<size_t>(slot) >= slot_events_.size() || !slot_pending_[static_cast<size_t>(slot)]) return;
  DeviceGuard g(device_);
  hipStream_t cur = current_stream(device_);
  hipEvent_t w0 = nullptr;
  if (timing_) {  // how long the compute stream is held up by this wait
    w0 = timing_event();
    DPA_HIP_CHECK(hipEventRecord(w0, cur));
  }
  DPA_HIP_CHECK(hipStreamWaitEvent(cur, slot_events_[static_cast<size_t>(slot)], 0));
  slot_pending_[static_cast<size_t>(slot)] = 0;
  if (w0) {
    hipEvent_t w1 = timing_event();
    DPA_HIP_CHECK(hipEventRecord(w1, cur));
    wait_spans_.emplace_back(w0, w1);
  }
}

void RcclComm::wait_all_slots() {
  for (size_t i = 0; i < slot_events_.size(); ++i) wait_slot(static_cast<int>(i));
}

void RcclComm::set_emulation(int ranks, double busbw_gbps, int workgroups, int passes, double latency_us) {
  if (ranks > 1 && world_ != 1) throw std::runtime_error("RcclComm: the all-reduce stand-in needs a one-rank communicator");
  if (ranks > 1 && !(busbw_gbps > 0.0)) throw std::runtime_error("RcclComm: emulated bus bandwidth must be positive");
  emu_ranks_ = ranks;
  emu_busbw_gbps_ = busbw_gbps;
  emu_workgroups_ = workgroups > 0 ? workgroups : 32;
  emu_passes_ = passes > 0 ? passes : 2;
  emu_latency_us_ = latency_us;
}

void RcclComm::set_timing(bool on) {
  timing_ = on;
  if (!on) {
    op_spans_.clear();
    wait_spans_.clear();
    timing_used_ = 0;
  }
}

std::vector<double> RcclComm::read_timing() {
  DeviceGuard g(device_);
  double busy = 0.0, exposed = 0.0;
  const double n_ops = static_cast<double>(op_spans_.size());
  // Blocking waits (hipEventSynchronize), never a polling loop over hipEventQuery.
  for (auto& sp : op_spans_) {
    DPA_HIP_CHECK(hipEventSynchronize(sp.second));
    float ms = 0.f;
    DPA_HIP_CHECK(hipEventElapsedTime(&ms, sp.first, sp.second));
    busy += ms;
  }
  for (auto& sp : wait_spans_) {
    DPA_HIP_CHECK(hipEventSynchronize(sp.second));
    float ms = 0.f;
    DPA_HIP_CHECK(hipEventElapsedTime(&ms, sp.first, sp.second));
    exposed += ms;
  }
  op_spans_.clear();
  wait_spans_.clear();
  timing_used_ = 0;
  return {busy, exposed, n_ops};
}

bool RcclComm::comm_stream_idle() {
  DeviceGuard g(device_);
  if (!comm_stream_) return true;
  return hipStreamQuery(comm_stream_) == hipSuccess;
}

void RcclComm::synchronize_comm_stream() {
  DeviceGuard g(device_);
  if (comm_stream_) DPA_HIP_CHECK(hipStreamSynchronize(comm_stream_));
}

void RcclComm::group_start() { DPA_NCCL_CHECK(ncclGroupStart()); }
void RcclComm::group_end() { DPA_NCCL_CHECK(ncclGroupEnd()); }

}  // namespace dpa
