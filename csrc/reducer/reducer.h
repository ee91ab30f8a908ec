// Gradient reducer: bucketed, overlapped all-reduce driven from autograd hooks.
//
// Owns what torch's C++ Reducer does for the DistributedDataParallel wrapper the reference
// builds in prepare_ddp_model (reference distributed.py:112-115), re-designed around flat
// arenas:
//   * every bucket is a slice of one flat buffer laid out by the planner (bucket_plan.h);
//   * a grad-accumulator post hook per parameter counts the bucket down; a complete bucket
//     is packed by ONE multi-tensor HIP launch (flatten + pre-divide + dtype cast) and its
//     all-reduce is queued on the communication stream while backward keeps running;
//   * buckets are always launched in index order, so every rank issues the same sequence
//     of collectives even when gradients become ready in a different order;
//   * with gradient-as-bucket-view the parameter's .grad is re-pointed at its slice, so no
//     unflatten copy exists and a fused optimizer can sweep the arena in one launch;
//   * an end-of-backward callback launches buckets that never completed (parameters that
//     received no gradient contribute zeros) and makes the compute stream wait for the
//     communication stream before the optimizer may run.
// The data plane is abstract (`Transport`): RCCL on the GPU, a c10d group (gloo) on the CPU.
#pragma once

#include <ATen/ATen.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <memory>
#include <mutex>
#include <vector>

#include "comm/rccl_comm.h"

namespace dpa {

class Transport {
 public:
  virtual ~Transport() = default;
  // True when the transport averages by itself (ncclAvg); otherwise the reducer pre-divides.
  virtual bool native_average() const = 0;
  virtual void all_reduce_async(at::Tensor flat, int slot, bool average) = 0;
  virtual void broadcast(at::Tensor flat, int root) = 0;
  // Sharded mode, both in place on a flat buffer of world * shard elements: after reduce_scatter rank r's shard
  // [r*shard, (r+1)*shard) holds the reduced (averaged) values; all_gather publishes every rank's shard to all.
  virtual void reduce_scatter_async(at::Tensor flat, int slot, bool average) = 0;
  virtual void all_gather_async(at::Tensor flat, int slot) = 0;
  virtual int rank() const = 0;
  // Blocking sum all-reduce of a small tensor (the "gradient used anywhere" bitmap).
  virtual void all_reduce_sum_blocking(at::Tensor t) = 0;
  // Make the consumer of the buckets (current stream / host thread) wait for all pending ops.
  virtual void wait_all() = 0;
  // Wait for one slot only; transports that cannot do that return false (the reducer then never defers).
  virtual bool wait_slot(int /*slot*/) { return false; }
  virtual bool per_slot_waits() const { return false; }
  virtual int world() const = 0;
};

std::shared_ptr<Transport> make_rccl_transport(std::shared_ptr<RcclComm> comm);
std::shared_ptr<Transport> make_c10d_transport(c10::intrusive_ptr<c10d::ProcessGroup> pg);
// Single-process transport (world 1): no communication, used by tests and to exercise the
// bucket life cycle on one GPU.
std::shared_ptr<Transport> make_local_transport();

struct BucketSpec {
  std::vector<int64_t> param_indices;
  std::vector<int64_t> offsets;  // element offsets inside the bucket buffer
  at::Tensor buffer;             // flat [numel] tensor in the reduce dtype
};

struct ReducerOptions {
  bool gradient_as_bucket_view = true;
  bool unpack_to_grads = true;  // when .grad is not a view: copy reduced values back into .grad
  // A bucket larger than this goes out as several all-reduces over consecutive slices of its
  // flat buffer (0 = one collective per bucket).  Keeps a single huge tensor (an embedding
  // table, an 8192x8192 layer) from being one monolithic operation: each slice has its own
  // completion event, and on the host-side (gloo) transport the slices run concurrently.
  int64_t max_collective_bytes = 0;
  // find_unused_parameters: after each synchronised backward the ranks agree (one small all-reduce and a
  // device-to-host copy, as in torch's reducer) on which parameters produced a gradient on ANY rank; a
  // parameter unused everywhere keeps grad = None instead of receiving zeros, so an optimizer with weight
  // decay or momentum leaves it alone.
  bool track_unused = false;
  // Sharded gradients: every bucket goes out as ONE in-place reduce-scatter (its element count must be a multiple
  // of the world size), after which this rank holds the averaged gradient of its own 1/world range of every bucket
  // only.  The consumer is a sharded optimizer that updates that range and all-gathers the parameters
  // (parallel/zero.py); .grad tensors outside the rank's range hold local, unreduced values.
  bool shard_gradients = false;
};

class Reducer {
 public:
  Reducer(std::vector<at::Tensor> params, std::vector<BucketSpec> buckets,
          std::shared_ptr<Transport> transport, ReducerOptions options);
  ~Reducer();

  // Arms the hooks for the backward pass that follows.  sync=false: gradients only
  // accumulate locally (no_sync()).
  void prepare_for_backward(bool sync);
  bool armed() const { return armed_; }
  // Parameter indices in the order their gradients became ready in the last synced backward.
  std::vector<int64_t> ready_order() const { return last_ready_order_; }
  int64_t num_buckets() const { return static_cast<int64_t>(buckets_.size()); }
  // Counters for tests / tracing.
  int64_t num_backward_syncs() const { return n_syncs_; }
  int64_t num_bucket_launches() const { return n_launches_; }
  int64_t num_pack_launches() const { return n_pack_launches_; }
  int64_t num_collectives() const { return n_collectives_; }
  // Per bucket: how many gradients had been reported ready when it was launched in the last
  // synced backward (a bucket laid out in ready order launches exactly when its last tensor arrives).
  std::vector<int64_t> launch_positions() const { return last_launch_pos_; }
  // Per bucket: number of all-reduce slices.
  std::vector<int64_t> slices_per_bucket() const;
  // Parameters that no rank produced a gradient for in the last synchronised backward (track_unused only).
  std::vector<int64_t> globally_unused() const { return last_globally_unused_; }
  // A fresh view of parameter `param_index`'s slice of its bucket, for an operator that wants to write the
  // parameter's gradient straight into the arena and return that view as the gradient (autograd then adopts it
  // as .grad and the bucket needs no pack copy).  Handed out at most once per backward pass and only while the
  // parameter has no .grad (an existing gradient must be accumulated into, not overwritten); undefined tensor
  // otherwise -- the caller allocates as usual.
  at::Tensor take_grad_sink(int64_t param_index);
  // Gradients found in place in their bucket / copied into it at bucket launch (since construction).
  int64_t num_grads_in_place() const { return n_in_place_; }
  int64_t num_grads_packed() const { return n_packed_; }
  // Broadcast every bucket-ordered parameter from `root` (initial parameter sync).
  void broadcast_parameters(int root);
  // Deferred waits.  Normally the end-of-backward callback makes the compute stream wait for every bucket's
  // all-reduce, so any kernel queued after backward() sees reduced gradients.  With deferral on, that wait is
  // left to a consumer that knows the bucket layout (the fused optimizers): it calls wait_bucket(b) right before
  // it touches bucket b's gradients, so the last buckets' all-reduces run under the update of the first ones.
  // Only effective when every gradient is a bucket view (nothing to unpack) and the transport has per-slot
  // events; anything still pending is waited for at the next prepare_for_backward().
  void set_deferred_wait(bool on) { defer_wait_ = on; }
  bool deferred_wait() const { return defer_wait_; }
  bool waits_pending() const { return waits_pending_; }
  void wait_bucket(int64_t b);
  void wait_pending();
  // Bucket index of a parameter (-1: none).
  int64_t bucket_of(int64_t param_index) const {
    return param_index >= 0 && param_index < static_cast<int64_t>(locator_.size()) ? locator_[static_cast<size_t>(param_index)].first : -1;
  }
  // Sharded mode: in-place all-gather of a flat parameter buffer laid out like bucket b, on the bucket's slot.
  void all_gather_bucket(int64_t b, at::Tensor flat);
  // Removes the autograd hooks (also done by the destructor).
  void detach();

 private:
  struct BucketState {
    BucketSpec spec;
    std::vector<at::Tensor> views;      // per tensor: buffer slice shaped like the parameter
    int64_t pending = 0;
    bool launched = false;
    std::vector<char> ready;            // per tensor in the bucket
    std::vector<std::pair<int64_t, int64_t>> slices;  // (element offset, numel) per collective
    int slot_base = 0;                  // first completion slot of this bucket
  };

  void autograd_hook(int64_t param_index);
  void launch_bucket(size_t b);
  void launch_ready_buckets();
  void finalize_backward();
  void pack_into_bucket(BucketState& bs, const std::vector<int64_t>& local_ids, float scale);
  void unpack_from_bucket(BucketState& bs);
  // dst[i] = cast(src[i] * scale) for every pair; dense GPU tensors of one (src, dst) dtype
  // combination share one multi-tensor launch, anything else falls back to copy_().
  void multi_copy(const std::vector<std::pair<at::Tensor, at::Tensor>>& src_dst, float scale);

  std::vector<at::Tensor> params_;
  std::vector<BucketState> buckets_;
  std::vector<std::pair<int64_t, int64_t>> locator_;  // param -> (bucket, index inside bucket)
  std::shared_ptr<Transport> transport_;
  ReducerOptions opt_;
  std::vector<std::pair<uintptr_t, std::shared_ptr<torch::autograd::Node>>> hooks_;

  std::mutex mu_;
  bool armed_ = false;
  bool callback_queued_ = false;
  size_t next_bucket_ = 0;
  std::vector<int64_t> ready_order_, last_ready_order_;
  std::vector<int64_t> launch_pos_, last_launch_pos_, last_globally_unused_;
  std::vector<char> had_grad_;  // per parameter: a gradient existed locally when its bucket was launched
  bool defer_wait_ = false, waits_pending_ = false;
  std::vector<char> bucket_waited_;
  std::vector<char> sink_taken_;  // per parameter: its bucket slice was handed out as a gradient sink in this pass
  int64_t n_syncs_ = 0, n_launches_ = 0, n_pack_launches_ = 0, n_collectives_ = 0, n_in_place_ = 0, n_packed_ = 0;
};

}  // namespace dpa
