#include "reducer.h"

#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/utils/lambda_post_hook.h>
#include <torch/csrc/autograd/variable.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "kernels/kernels.h"
#include "trace.h"

namespace dpa {

namespace {

int kernel_dtype(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return kernels::kF32;
    case at::kBFloat16: return kernels::kBF16;
    case at::kHalf: return kernels::kF16;
    default: return -1;
  }
}

hipStream_t current_stream_of(const at::Tensor& t) {
  return c10::hip::getCurrentHIPStream(static_cast<c10::DeviceIndex>(t.get_device())).stream();
}

// Appends (src -> dst) to a copy table, flushing it through `flush` when it is full.
template <typename Flush>
void table_add(kernels::CopyTable& tab, const void* src, void* dst, int64_t numel, Flush&& flush) {
  if (numel <= 0) return;
  if (tab.n == kernels::CopyTable::kMaxTensors) {
    flush(tab);
    tab.n = 0;
    tab.block_start[0] = 0;
  }
  const int i = tab.n;
  tab.src[i] = src;
  tab.dst[i] = dst;
  tab.numel[i] = numel;
  const int blocks = static_cast<int>((numel + kernels::CopyTable::kChunk - 1) / kernels::CopyTable::kChunk);
  tab.block_start[i + 1] = tab.block_start[i] + blocks;
  tab.n = i + 1;
}

// ------------------------------------------------------------------ transports
class RcclTransport final : public Transport {
 public:
  explicit RcclTransport(std::shared_ptr<RcclComm> c) : comm_(std::move(c)) {}
  bool native_average() const override { return true; }
  void all_reduce_async(at::Tensor flat, int slot, bool average) override {
    comm_->all_reduce_async(flat, average ? static_cast<int>(ncclAvg) : static_cast<int>(ncclSum), slot);
  }
  void broadcast(at::Tensor flat, int root) override { comm_->broadcast(flat, root); }
  void reduce_scatter_async(at::Tensor flat, int slot, bool average) override {
    comm_->reduce_scatter_async(flat, average ? static_cast<int>(ncclAvg) : static_cast<int>(ncclSum), slot);
  }
  void all_gather_async(at::Tensor flat, int slot) override { comm_->all_gather_async(flat, slot); }
  int rank() const override { return comm_->rank(); }
  void all_reduce_sum_blocking(at::Tensor t) override {
    comm_->all_reduce(t, static_cast<int>(ncclSum));  // on the current stream; the caller's copy to the host waits for it
  }
  void wait_all() override { comm_->wait_all_slots(); }
  bool wait_slot(int slot) override {
    comm_->wait_slot(slot);
    return true;
  }
  bool per_slot_waits() const override { return true; }
  int world() const override { return comm_->world(); }

 private:
  std::shared_ptr<RcclComm> comm_;
};

class C10dTransport final : public Transport {
 public:
  explicit C10dTransport(c10::intrusive_ptr<c10d::ProcessGroup> pg) : pg_(std::move(pg)) {
    const char* e = std::getenv("DIST_C10D_TRACE");
    trace_ = e != nullptr && e[0] == '1';
  }
  // DIST_C10D_TRACE=1: one stderr line per collective this transport enters (rank, operation, elements), for
  // comparing the ranks' sequences when a host-side run hangs.
  void note(const char* what, int64_t numel) const {
    if (trace_) std::fprintf(stderr, "[c10d-transport %p rank %d] %s numel=%lld pending=%zu\n", static_cast<const void*>(this), pg_->getRank(), what,
                             static_cast<long long>(numel), works_.size());
  }
  bool native_average() const override { return false; }
  // GPU tensors on a host-side group (gloo) are staged through host memory HERE, with plain synchronous copies on
  // the current stream, and the group only ever sees CPU tensors.  Handing it the GPU tensor makes the group stage
  // it itself on HIGH-PRIORITY streams of torch's pool, with event waits between those streams and the compute
  // stream in both directions.  A high-priority hardware queue that shares a command-processor pipe with the
  // compute stream's queue keeps that pipe while it waits, so the compute queue can never signal the event it
  // waits for (comm/comm_stream.h); which queue lands on which pipe depends on how many queues the process made
  // before, hence "always with one model, every other run with another".  This path is a plumbing
  // configuration (several ranks sharing one GPU in tests), not a fast one, so nothing is overlapped.
  static void reduce_through_host(c10d::ProcessGroup& pg, at::Tensor& t) {
    at::Tensor h = t.to(at::kCPU);  // device-to-host copy; returns when the data is there
    std::vector<at::Tensor> ts{h};
    c10d::AllreduceOptions o;
    o.reduceOp = c10d::ReduceOp::SUM;
    pg.allreduce(ts, o)->wait();
    t.copy_(h);
  }
  void all_reduce_async(at::Tensor flat, int /*slot*/, bool /*average: pre-divided by the reducer*/) override {
    note("all_reduce_async", flat.numel());
    if (flat.is_cuda()) {
      reduce_through_host(*pg_, flat);
      return;
    }
    std::vector<at::Tensor> ts{flat};
    c10d::AllreduceOptions o;
    o.reduceOp = c10d::ReduceOp::SUM;
    works_.push_back(pg_->allreduce(ts, o));
  }
  void broadcast(at::Tensor flat, int root) override {
    note("broadcast", flat.numel());
    at::Tensor h = flat.is_cuda() ? flat.to(at::kCPU) : flat;
    std::vector<at::Tensor> ts{h};
    c10d::BroadcastOptions o;
    o.rootRank = root;
    pg_->broadcast(ts, o)->wait();
    if (flat.is_cuda() && pg_->getRank() != root) flat.copy_(h);
  }
  // The host-side group is the correctness path: a reduce-scatter is served by the all-reduce (every rank then holds
  // every range, a superset of what the caller may rely on).
  void reduce_scatter_async(at::Tensor flat, int slot, bool average) override { all_reduce_async(flat, slot, average); }
  void all_gather_async(at::Tensor flat, int /*slot*/) override {
    note("all_gather", flat.numel());
    wait_all();
    const int64_t n = pg_->getSize(), shard = flat.numel() / n;
    at::Tensor h = flat.is_cuda() ? flat.to(at::kCPU) : flat;
    // one all-gather: the rank's own range goes in as a copy, the ranges of the flat buffer are the outputs
    std::vector<at::Tensor> in{h.narrow(0, pg_->getRank() * shard, shard).clone()};
    std::vector<std::vector<at::Tensor>> out(1);
    for (int64_t r = 0; r < n; ++r) out[0].push_back(h.narrow(0, r * shard, shard));
    pg_->allgather(out, in)->wait();
    if (flat.is_cuda()) flat.copy_(h);
  }
  int rank() const override { return pg_->getRank(); }
  void all_reduce_sum_blocking(at::Tensor t) override {
    if (t.is_cuda()) {
      reduce_through_host(*pg_, t);
      return;
    }
    std::vector<at::Tensor> ts{t};
    c10d::AllreduceOptions o;
    o.reduceOp = c10d::ReduceOp::SUM;
    pg_->allreduce(ts, o)->wait();
  }
  void wait_all() override {
    for (auto& w : works_) w->wait();
    works_.clear();
  }
  int world() const override { return pg_->getSize(); }

 private:
  c10::intrusive_ptr<c10d::ProcessGroup> pg_;
  std::vector<c10::intrusive_ptr<c10d::Work>> works_;
  bool trace_ = false;
};

class LocalTransport final : public Transport {
 public:
  bool native_average() const override { return true; }
  void all_reduce_async(at::Tensor, int, bool) override {}
  void reduce_scatter_async(at::Tensor, int, bool) override {}
  void all_gather_async(at::Tensor, int) override {}
  int rank() const override { return 0; }
  void broadcast(at::Tensor, int) override {}
  void all_reduce_sum_blocking(at::Tensor) override {}
  void wait_all() override {}
  int world() const override { return 1; }
};

}  // namespace

std::shared_ptr<Transport> make_rccl_transport(std::shared_ptr<RcclComm> comm) {
  return std::make_shared<RcclTransport>(std::move(comm));
}
std::shared_ptr<Transport> make_c10d_transport(c10::intrusive_ptr<c10d::ProcessGroup> pg) {
  return std::make_shared<C10dTransport>(std::move(pg));
}
std::shared_ptr<Transport> make_local_transport() { return std::make_shared<LocalTransport>(); }

// ------------------------------------------------------------------ reducer
Reducer::Reducer(std::vector<at::Tensor> params, std::vector<BucketSpec> buckets, std::shared_ptr<Transport> transport,
                 ReducerOptions options)
    : params_(std::move(params)), transport_(std::move(transport)), opt_(options) {
  locator_.assign(params_.size(), {-1, -1});
  buckets_.reserve(buckets.size());
  int next_slot = 0;
  for (size_t b = 0; b < buckets.size(); ++b) {
    BucketState bs;
    bs.spec = std::move(buckets[b]);
    const auto& sp = bs.spec;
    if (sp.param_indices.size() != sp.offsets.size()) throw std::invalid_argument("Reducer: indices/offsets size mismatch");
    if (!sp.buffer.defined() || sp.buffer.dim() != 1 || !sp.buffer.is_contiguous())
      throw std::invalid_argument("Reducer: bucket buffer must be a contiguous 1-D tensor");
    for (size_t j = 0; j < sp.param_indices.size(); ++j) {
      const int64_t pi = sp.param_indices[j];
      if (pi < 0 || pi >= static_cast<int64_t>(params_.size())) throw std::out_of_range("Reducer: parameter index out of range");
      if (locator_[static_cast<size_t>(pi)].first >= 0) throw std::invalid_argument("Reducer: parameter assigned to two buckets");
      const at::Tensor& p = params_[static_cast<size_t>(pi)];
      if (p.device() != sp.buffer.device()) throw std::invalid_argument("Reducer: parameter and bucket on different devices");
      if (sp.offsets[j] < 0 || sp.offsets[j] + p.numel() > sp.buffer.numel())
        throw std::out_of_range("Reducer: tensor does not fit its bucket slice");
      locator_[static_cast<size_t>(pi)] = {static_cast<int64_t>(b), static_cast<int64_t>(j)};
      bs.views.push_back(sp.buffer.narrow(0, sp.offsets[j], p.numel()).view(p.sizes()));
    }
    bs.ready.assign(sp.param_indices.size(), 0);
    bs.pending = static_cast<int64_t>(sp.param_indices.size());
    // Collective slices: whole bucket, or cap-sized pieces whose boundaries stay 128-byte aligned.
    const int64_t total = sp.buffer.numel();
    const int64_t esz = static_cast<int64_t>(sp.buffer.element_size());
    int64_t per = total;
    if (opt_.max_collective_bytes > 0 && total * esz > opt_.max_collective_bytes) {
      per = std::max<int64_t>(opt_.max_collective_bytes / esz / 128 * 128, 128);
    }
    for (int64_t off = 0; off < total || (total == 0 && off == 0); off += per) {
      bs.slices.emplace_back(off, std::min(per, total - off));
      if (total == 0) break;
    }
    if (opt_.shard_gradients) {
      if (total % transport_->world() != 0) throw std::invalid_argument("Reducer: a sharded bucket must hold a multiple of world-size elements");
      bs.slices.assign(1, {0, total});
    }
    bs.slot_base = next_slot;
    next_slot += static_cast<int>(bs.slices.size());
    buckets_.push_back(std::move(bs));
  }
  for (size_t i = 0; i < params_.size(); ++i) {
    if (!params_[i].requires_grad()) continue;
    if (locator_[i].first < 0) throw std::invalid_argument("Reducer: a parameter that requires grad is in no bucket");
    auto acc = torch::autograd::impl::grad_accumulator(params_[i]);
    const int64_t idx = static_cast<int64_t>(i);
    uintptr_t key = acc->add_post_hook(std::make_unique<torch::autograd::utils::LambdaPostHook>(
        [this, idx](const torch::autograd::variable_list& outputs, const torch::autograd::variable_list& /*inputs*/) {
          this->autograd_hook(idx);
          return outputs;
        }));
    hooks_.emplace_back(key, std::move(acc));
  }
}

Reducer::~Reducer() { detach(); }

void Reducer::detach() {
  for (auto& h : hooks_) h.second->del_post_hook(h.first);
  hooks_.clear();
}

void Reducer::wait_bucket(int64_t b) {
  std::lock_guard<std::mutex> g(mu_);
  if (!waits_pending_ || b < 0 || b >= static_cast<int64_t>(buckets_.size()) || bucket_waited_[static_cast<size_t>(b)]) return;
  const BucketState& bs = buckets_[static_cast<size_t>(b)];
  for (size_t k = 0; k < bs.slices.size(); ++k) transport_->wait_slot(bs.slot_base + static_cast<int>(k));
  bucket_waited_[static_cast<size_t>(b)] = 1;
}

void Reducer::wait_pending() {
  std::lock_guard<std::mutex> g(mu_);
  if (!waits_pending_) return;
  transport_->wait_all();
  waits_pending_ = false;
}

void Reducer::prepare_for_backward(bool sync) {
  std::lock_guard<std::mutex> g(mu_);
  if (waits_pending_) {  // nobody consumed the deferred waits of the previous pass
    transport_->wait_all();
    waits_pending_ = false;
  }
  armed_ = sync;
  callback_queued_ = false;
  next_bucket_ = 0;
  ready_order_.clear();
  launch_pos_.assign(buckets_.size(), -1);
  had_grad_.assign(params_.size(), 0);
  sink_taken_.assign(params_.size(), 0);
  for (auto& bs : buckets_) {
    bs.pending = static_cast<int64_t>(bs.spec.param_indices.size());
    bs.launched = false;
    std::fill(bs.ready.begin(), bs.ready.end(), 0);
  }
}

at::Tensor Reducer::take_grad_sink(int64_t param_index) {
  std::lock_guard<std::mutex> g(mu_);
  if (!opt_.gradient_as_bucket_view || param_index < 0 || param_index >= static_cast<int64_t>(params_.size())) return at::Tensor();
  const size_t i = static_cast<size_t>(param_index);
  const auto loc = locator_[i];
  if (loc.first < 0 || sink_taken_.size() != params_.size() || sink_taken_[i]) return at::Tensor();
  const at::Tensor& p = params_[i];
  BucketState& bs = buckets_[static_cast<size_t>(loc.first)];
  if (p.grad().defined() || bs.launched || bs.spec.buffer.scalar_type() != p.scalar_type() || !p.is_contiguous()) return at::Tensor();
  sink_taken_[i] = 1;
  const size_t j = static_cast<size_t>(loc.second);
  return bs.spec.buffer.narrow(0, bs.spec.offsets[j], p.numel()).view(p.sizes());
}

void Reducer::autograd_hook(int64_t param_index) {
  std::lock_guard<std::mutex> g(mu_);
  if (!armed_) return;
  if (!callback_queued_) {
    callback_queued_ = true;
    torch::autograd::Engine::get_default_engine().queue_callback([this] { this->finalize_backward(); });
  }
  const auto loc = locator_[static_cast<size_t>(param_index)];
  BucketState& bs = buckets_[static_cast<size_t>(loc.first)];
  if (bs.launched)
    throw std::runtime_error(
        "Reducer: a parameter produced a second gradient after its bucket was already reduced in this backward pass "
        "(parameter used in several backward graphs: wrap the extra passes in no_sync())");
  if (bs.ready[static_cast<size_t>(loc.second)]) return;  // same parameter hit twice before launch: grad already accumulated
  bs.ready[static_cast<size_t>(loc.second)] = 1;
  ready_order_.push_back(param_index);
  if (--bs.pending == 0) launch_ready_buckets();
}

void Reducer::launch_ready_buckets() {
  while (next_bucket_ < buckets_.size() && buckets_[next_bucket_].pending == 0) {
    launch_bucket(next_bucket_);
    ++next_bucket_;
  }
}

std::vector<int64_t> Reducer::slices_per_bucket() const {
  std::vector<int64_t> out;
  for (const auto& bs : buckets_) out.push_back(static_cast<int64_t>(bs.slices.size()));
  return out;
}

void Reducer::multi_copy(const std::vector<std::pair<at::Tensor, at::Tensor>>& src_dst, float scale) {
  if (src_dst.empty()) return;
  at::NoGradGuard ng;
  auto fast = [](const at::Tensor& t) { return t.is_cuda() && t.is_contiguous() && kernel_dtype(t.scalar_type()) >= 0; };
  std::vector<size_t> fast_ids;
  for (size_t i = 0; i < src_dst.size(); ++i) {
    const at::Tensor& src = src_dst[i].first;
    at::Tensor dst = src_dst[i].second;
    if (src.numel() != dst.numel()) throw std::invalid_argument("Reducer: copy between tensors of different size");
    if (fast(src) && fast(dst) && src.get_device() == dst.get_device()) {
      fast_ids.push_back(i);
    } else {
      dst.copy_(src);
      if (scale != 1.f) dst.mul_(scale);
    }
  }
  if (fast_ids.empty()) return;
  hipStream_t stream = current_stream_of(src_dst[fast_ids[0]].second);
  for (int src_dt = 0; src_dt < 3; ++src_dt)
    for (int dst_dt = 0; dst_dt < 3; ++dst_dt) {
      kernels::CopyTable tab;
      tab.n = 0;
      tab.block_start[0] = 0;
      auto flush = [&](kernels::CopyTable& t) {
        kernels::launch_multi_copy_scale(t, src_dt, dst_dt, scale, stream);
        ++n_pack_launches_;
      };
      for (size_t i : fast_ids) {
        const at::Tensor& src = src_dst[i].first;
        const at::Tensor& dst = src_dst[i].second;
        if (kernel_dtype(src.scalar_type()) != src_dt || kernel_dtype(dst.scalar_type()) != dst_dt) continue;
        table_add(tab, src.data_ptr(), dst.data_ptr(), src.numel(), flush);
      }
      if (tab.n > 0) flush(tab);
    }
}

void Reducer::pack_into_bucket(BucketState& bs, const std::vector<int64_t>& local_ids, float scale) {
  std::vector<std::pair<at::Tensor, at::Tensor>> pairs;
  pairs.reserve(local_ids.size());
  for (int64_t j : local_ids)
    pairs.emplace_back(params_[static_cast<size_t>(bs.spec.param_indices[static_cast<size_t>(j)])].grad(),
                       bs.views[static_cast<size_t>(j)]);
  multi_copy(pairs, scale);
}

void Reducer::unpack_from_bucket(BucketState& bs) {
  std::vector<std::pair<at::Tensor, at::Tensor>> pairs;
  for (size_t j = 0; j < bs.spec.param_indices.size(); ++j) {
    at::Tensor& p = params_[static_cast<size_t>(bs.spec.param_indices[j])];
    const at::Tensor& g = p.grad();
    if (!g.defined()) {
      // Unused on every rank (known only with track_unused): grad stays None.
      if (opt_.track_unused &&
          std::binary_search(last_globally_unused_.begin(), last_globally_unused_.end(), bs.spec.param_indices[j]))
        continue;
      // Locally unused parameter: it still receives the globally reduced gradient.
      p.mutable_grad() = bs.views[j].to(p.scalar_type());
      continue;
    }
    if (g.data_ptr() == bs.views[j].data_ptr()) continue;
    pairs.emplace_back(bs.views[j], g);
  }
  multi_copy(pairs, 1.f);
}

void Reducer::launch_bucket(size_t b) {
  BucketState& bs = buckets_[b];
  trace::Range range(trace::enabled() ? "bucket " + std::to_string(b) + " launch (" +
                                            std::to_string(bs.spec.buffer.numel() * bs.spec.buffer.element_size()) + " B)"
                                      : std::string());
  const bool pre_divide = !transport_->native_average() && transport_->world() > 1;
  const float scale = pre_divide ? 1.f / static_cast<float>(transport_->world()) : 1.f;
  std::vector<int64_t> to_copy;
  for (size_t j = 0; j < bs.spec.param_indices.size(); ++j) {
    at::Tensor& p = params_[static_cast<size_t>(bs.spec.param_indices[j])];
    const at::Tensor& g = p.grad();
    at::Tensor& v = bs.views[j];
    had_grad_[static_cast<size_t>(bs.spec.param_indices[j])] = g.defined() ? 1 : 0;
    if (!g.defined()) {
      v.zero_();  // no gradient exists locally: this rank contributes zeros
    } else if (g.data_ptr() == v.data_ptr()) {
      // Accumulated in place inside the bucket (also a gradient kept from earlier no_sync()
      // micro-steps of a parameter that was not used in this one).
      if (pre_divide) v.mul_(scale);
      ++n_in_place_;
    } else {
      // A gradient that lives outside the arena, produced now or accumulated under no_sync().
      to_copy.push_back(static_cast<int64_t>(j));
      ++n_packed_;
    }
  }
  pack_into_bucket(bs, to_copy, scale);
  if (opt_.gradient_as_bucket_view) {
    for (int64_t j : to_copy) {
      at::Tensor& p = params_[static_cast<size_t>(bs.spec.param_indices[static_cast<size_t>(j)])];
      if (p.scalar_type() == bs.spec.buffer.scalar_type()) p.mutable_grad() = bs.views[static_cast<size_t>(j)];
    }
  }
  for (size_t k = 0; k < bs.slices.size(); ++k) {
    const auto& sl = bs.slices[k];
    at::Tensor piece = bs.slices.size() == 1 ? bs.spec.buffer : bs.spec.buffer.narrow(0, sl.first, sl.second);
    if (opt_.shard_gradients) transport_->reduce_scatter_async(piece, bs.slot_base + static_cast<int>(k), /*average=*/true);
    else transport_->all_reduce_async(piece, bs.slot_base + static_cast<int>(k), /*average=*/true);
    ++n_collectives_;
  }
  bs.launched = true;
  launch_pos_[b] = static_cast<int64_t>(ready_order_.size());
  ++n_launches_;
}

void Reducer::finalize_backward() {
  std::lock_guard<std::mutex> g(mu_);
  if (!armed_) return;
  trace::Range range("reducer finalize");
  // Buckets that never filled up (parameters without a gradient this iteration) go out now,
  // still in index order, so the collective sequence is identical on every rank.
  for (size_t b = next_bucket_; b < buckets_.size(); ++b) launch_bucket(b);
  next_bucket_ = buckets_.size();
  bool all_views = opt_.gradient_as_bucket_view && !opt_.track_unused;
  if (all_views && defer_wait_ && transport_->per_slot_waits()) {
    for (auto& bs : buckets_) {
      for (size_t j = 0; j < bs.spec.param_indices.size() && all_views; ++j) {
        const at::Tensor& gr = params_[static_cast<size_t>(bs.spec.param_indices[j])].grad();
        all_views = gr.defined() && gr.data_ptr() == bs.views[j].data_ptr();
      }
      if (!all_views) break;
    }
  } else {
    all_views = false;
  }
  if (all_views) {
    waits_pending_ = true;
    bucket_waited_.assign(buckets_.size(), 0);
  } else {
    transport_->wait_all();
  }
  last_globally_unused_.clear();
  if (opt_.track_unused) {
    // Which parameters got a gradient on at least one rank: sum of the local 0/1 flags.
    at::Tensor flags = at::empty({static_cast<int64_t>(params_.size())}, at::TensorOptions().dtype(at::kInt));
    int32_t* fp = flags.data_ptr<int32_t>();
    for (size_t i = 0; i < params_.size(); ++i) fp[i] = had_grad_[i];
    const at::Device dev = buckets_.empty() ? at::Device(at::kCPU) : buckets_[0].spec.buffer.device();
    at::Tensor on_dev = dev.is_cpu() ? flags : flags.to(dev);
    transport_->all_reduce_sum_blocking(on_dev);
    at::Tensor back = on_dev.is_cpu() ? on_dev : on_dev.cpu();
    const int32_t* bp = back.data_ptr<int32_t>();
    for (size_t i = 0; i < params_.size(); ++i)
      if (locator_[i].first >= 0 && bp[i] == 0) last_globally_unused_.push_back(static_cast<int64_t>(i));
  }
  if (opt_.unpack_to_grads && !waits_pending_)
    for (auto& bs : buckets_) unpack_from_bucket(bs);
  armed_ = false;
  callback_queued_ = false;
  last_ready_order_ = ready_order_;
  last_launch_pos_ = launch_pos_;
  ++n_syncs_;
}

void Reducer::all_gather_bucket(int64_t b, at::Tensor flat) {
  std::lock_guard<std::mutex> g(mu_);
  if (b < 0 || b >= static_cast<int64_t>(buckets_.size())) throw std::out_of_range("Reducer: bucket index out of range");
  const BucketState& bs = buckets_[static_cast<size_t>(b)];
  if (flat.numel() != bs.spec.buffer.numel()) throw std::invalid_argument("Reducer: parameter buffer does not match the bucket layout");
  transport_->all_gather_async(flat, bs.slot_base);
  ++n_collectives_;
  waits_pending_ = true;  // the consumer (wait_bucket / wait_pending, or the next prepare_for_backward) orders the compute stream behind it
  if (bucket_waited_.size() != buckets_.size()) bucket_waited_.assign(buckets_.size(), 0);
  bucket_waited_[static_cast<size_t>(b)] = 0;
}

void Reducer::broadcast_parameters(int root) {
  // Parameters travel bucket by bucket through the (already allocated) gradient buffers:
  // one multi-tensor pack launch, one broadcast, one unpack launch per bucket, so no second
  // flat copy of the model is needed.
  at::NoGradGuard ng;
  for (auto& bs : buckets_) {
    std::vector<std::pair<at::Tensor, at::Tensor>> in, out;
    for (size_t j = 0; j < bs.spec.param_indices.size(); ++j) {
      at::Tensor p = params_[static_cast<size_t>(bs.spec.param_indices[j])].detach();
      // A bucket narrower than its parameters (bf16 buckets under fp32 parameters) would round
      // the weights on the way: such tensors are broadcast on their own.
      if (p.element_size() > bs.spec.buffer.element_size()) {
        at::Tensor flat = p.is_contiguous() ? p : p.contiguous();
        transport_->broadcast(flat.view({-1}), root);
        if (!p.is_contiguous()) p.copy_(flat);
        continue;
      }
      in.emplace_back(p, bs.views[j]);
      out.emplace_back(bs.views[j], p);
    }
    if (in.empty()) continue;
    multi_copy(in, 1.f);
    transport_->broadcast(bs.spec.buffer, root);
    multi_copy(out, 1.f);
    bs.spec.buffer.zero_();
  }
}

}  // namespace dpa
