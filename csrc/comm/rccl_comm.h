// RCCL communicator bound to one GPU, driven directly (no c10d ProcessGroup).
//
// This is the GPU data plane behind every collective of the `distributed` facade -- the
// role ProcessGroupNCCL plays for the reference (reference distributed.py:62-66, 119-177):
//   * the ncclUniqueId travels through our TCP store, the communicator is created eagerly;
//   * a dedicated HIP stream (one per process and device, see comm_stream.h for how it is chosen)
//     carries the gradient-bucket all-reduces so they overlap the backward kernels running on
//     torch's current stream; ordering between the two streams is expressed with hipEvents only
//     (no host synchronisation);
//   * metric-sized collectives (reduce / gather / barrier) are issued on the current stream,
//     where a dependency on the preceding kernels already exists, so they cost no fences;
//   * gather uses RCCL's native ncclGather instead of grouped send/recv.
// xGMI peer-to-peer is left enabled (the reference disables P2P, distributed.py:47).
#pragma once

#include <ATen/ATen.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "comm/comm_stream.h"

namespace dpa {

class RcclComm {
 public:
  static std::string unique_id();  // 128 raw bytes from ncclGetUniqueId
  // PCI bus id ("0000:05:00.0") of HIP device `device`: the identity ranks compare before
  // ncclCommInitRank so that two ranks bound to one GPU are reported instead of hanging.
  static std::string device_bus_id(int device);
  static std::string version();  // "2.26.6"-style RCCL version string

  // high_priority_stream: ask for a high-priority communication stream (only used when the probe of
  // comm_stream.h finds one that is independent of the compute stream; normal priority otherwise).
  RcclComm(const std::string& uid, int rank, int world, int device, bool high_priority_stream = false);
  ~RcclComm();
  RcclComm(const RcclComm&) = delete;
  RcclComm& operator=(const RcclComm&) = delete;

  int rank() const { return rank_; }
  int world() const { return world_; }
  int device() const { return device_; }

  // ---- collectives on torch's *current* stream (stream-ordered with the caller's kernels)
  void all_reduce(at::Tensor t, int op);
  void reduce(at::Tensor t, int root, int op);
  void broadcast(at::Tensor t, int root);
  void gather(at::Tensor in, c10::optional<at::Tensor> out, int root);
  void all_gather(at::Tensor in, at::Tensor out);
  void reduce_scatter(at::Tensor in, at::Tensor out, int op);
  void barrier();  // 1-element all-reduce, then the host waits for it

  // ---- collectives on the dedicated communication stream (gradient buckets)
  // The comm stream first waits for everything queued so far on the current stream, then
  // runs the all-reduce; `slot` names the completion event to wait on later.
  void all_reduce_async(at::Tensor t, int op, int slot);
  void broadcast_async(at::Tensor t, int root, int slot);
  // In-place sharded collectives on the communication stream: `t` is a flat buffer of world * shard elements.
  // reduce_scatter: this rank's shard [rank*shard, (rank+1)*shard) receives the reduction over ranks of that range
  // (the rest of `t` keeps local values); all_gather: every rank's shard is sent to its position in all ranks' `t`.
  void reduce_scatter_async(at::Tensor t, int op, int slot);
  void all_gather_async(at::Tensor t, int slot);
  void wait_slot(int slot);         // current stream waits for slot's completion event
  void wait_all_slots();            // ... for every slot used since the last call
  void synchronize_comm_stream();   // host waits (debug / teardown)
  hipStream_t comm_stream() {
    ensure_comm_stream();
    return comm_stream_;
  }
  // Use an existing stream (e.g. a torch.cuda.Stream created by the caller) for the async ops.
  void set_comm_stream(uintptr_t handle);
  // How the communication stream was chosen (priority, probe results).
  const CommStreamChoice& comm_stream_choice() const { return choice_; }

  void group_start();
  void group_end();

  void destroy();  // ncclCommFinalize + ncclCommDestroy; idempotent
  // ncclCommAbort: unblocks kernels of a collective a peer never joined (teardown on hang).  May be called
  // from another thread (heartbeat, signal watcher) while the training thread is enqueueing: enqueues hold
  // `comm_mu_` around the library call and give up once `aborted_` is set; abort() takes the same mutex, but
  // only waits a bounded time for it (a training thread blocked INSIDE the library is the case abort exists for).
  void abort();
  bool aborted() const { return aborted_.load(); }

  int64_t num_async_ops() const { return n_async_; }

  // Debug serialisation: every asynchronous collective is followed by a host wait for the
  // communication stream, so compute and communication never run concurrently.
  void set_debug_sync(bool on) { debug_sync_ = on; }
  bool debug_sync() const { return debug_sync_; }
  // True when nothing is queued on the communication stream (blocking check, no event polling).
  bool comm_stream_idle();

  // One-GPU stand-in for a multi-rank all-reduce (experiments only, never a multi-GPU result): while set, the
  // asynchronous all-reduces of a ONE-rank communicator launch, instead of the library call, a kernel on the
  // communication stream that keeps `workgroups` workgroups resident for the time an `ranks`-rank all-reduce
  // of the same bytes would take at `busbw_gbps` bus bandwidth (2 (n-1)/n * bytes / busbw + latency_us) and
  // moves the buffer `passes` times through HBM meanwhile.  Same fences, events and stream as the real path.
  void set_emulation(int ranks, double busbw_gbps, int workgroups, int passes, double latency_us);
  bool emulating() const { return emu_ranks_ > 1; }
  // A one-rank all-reduce is the identity; the library still runs a kernel that copies the buffer over itself.
  // With this on, the asynchronous all-reduces of a ONE-rank communicator skip the library call (fences and
  // completion events stay), which isolates what the reducer machinery itself costs (DIST_ONE_RANK_IDENTITY=1).
  void set_one_rank_identity(bool on) { one_rank_identity_ = on && world_ == 1; }
  bool one_rank_identity() const { return one_rank_identity_; }

  // Overlap accounting.  While enabled, every asynchronous collective is bracketed by two
  // timing events on the communication stream and wait_all_slots() by two on the compute
  // stream.  read_timing() waits for them and returns {comm_busy_ms, exposed_ms, n_ops} of
  // the window since the previous read: comm_busy_ms is the summed run time of the
  // collectives, exposed_ms the time the compute stream spent blocked on them (zero when the
  // backward pass hides the communication completely).
  void set_timing(bool on);
  std::vector<double> read_timing();

 private:
  void begin_slot(int slot);
  void ensure_comm_stream();
  void check_tensor(const at::Tensor& t) const;

  ncclComm_t comm_ = nullptr;
  std::timed_mutex comm_mu_;          // guards comm_ against abort() from another thread
  std::atomic<bool> aborted_{false};
  int rank_, world_, device_;
  bool want_high_ = false;
  hipStream_t comm_stream_ = nullptr;
  CommStreamChoice choice_;
  std::vector<hipEvent_t> slot_events_;   // completion of the operation on the comm stream
  std::vector<hipEvent_t> fence_events_;  // "inputs are ready" on the stream that produced them
  std::vector<char> slot_pending_;
  at::Tensor barrier_scratch_;
  int64_t n_async_ = 0;
  bool destroyed_ = false;
  bool debug_sync_ = false;
  bool timing_ = false;
  bool one_rank_identity_ = false;
  int emu_ranks_ = 0, emu_workgroups_ = 32, emu_passes_ = 2;
  double emu_busbw_gbps_ = 0.0, emu_latency_us_ = 30.0;
  std::vector<hipEvent_t> timing_pool_;                       // timing-enabled events, reused
  size_t timing_used_ = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> op_spans_;    // per collective, on the comm stream
  std::vector<std::pair<hipEvent_t, hipEvent_t>> wait_spans_;  // per wait_all_slots, on the compute stream
  hipEvent_t timing_event();
  void finish_async(int slot, hipEvent_t done, hipEvent_t t0);
};

ncclDataType_t to_nccl_dtype(at::ScalarType t);

}  // namespace dpa
