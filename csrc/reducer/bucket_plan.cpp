#include "bucket_plan.h"

#include <algorithm>
#include <map>
#include <sstream>
#include <stdexcept>

namespace dpa {

namespace {

int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

Plan plan_in_order(const std::vector<ParamMeta>& params, const std::vector<int64_t>& order,
                   const PlanOptions& opt) {
  if (opt.align_bytes <= 0 || opt.bucket_bytes <= 0 || opt.first_bucket_bytes <= 0)
    throw std::invalid_argument("plan_buckets: sizes must be positive");
  Plan plan;
  // One open bucket per group; a group's first bucket uses the small cap so the first
  // all-reduce starts early in backward, later ones the large cap.
  struct Open {
    Bucket b;
    int64_t bytes = 0;
    int n_closed = 0;
  };
  std::map<int32_t, Open> open;
  std::map<int32_t, int64_t> arena;

  auto close = [&](Open& o) {
    if (o.b.param_indices.empty()) return;
    o.b.arena_offset = arena[o.b.group_key];
    arena[o.b.group_key] += o.b.numel;
    plan.buckets.push_back(std::move(o.b));
    o.b = Bucket{};
    o.bytes = 0;
    o.n_closed += 1;
  };

  for (int64_t idx : order) {
    if (idx < 0 || idx >= static_cast<int64_t>(params.size()))
      throw std::out_of_range("plan_buckets: parameter index out of range");
    const ParamMeta& p = params[static_cast<size_t>(idx)];
    if (!p.requires_grad) continue;
    if (p.elem_size <= 0 || opt.align_bytes % p.elem_size != 0)
      throw std::invalid_argument("plan_buckets: align_bytes must be a multiple of elem_size");
    Open& o = open[p.group_key];
    if (o.b.param_indices.empty()) {
      o.b.group_key = p.group_key;
      o.b.elem_size = p.elem_size;
    }
    const int64_t align_elems = opt.align_bytes / p.elem_size;
    const int64_t off = round_up(o.b.numel, align_elems);
    o.b.param_indices.push_back(idx);
    o.b.offsets.push_back(off);
    o.b.numel = round_up(off + p.numel, align_elems);
    o.bytes = o.b.numel * p.elem_size;
    const int64_t cap = (o.n_closed == 0) ? opt.first_bucket_bytes : opt.bucket_bytes;
    if (o.bytes >= cap) close(o);
  }
  for (auto& kv : open) close(kv.second);
  for (auto& kv : arena) plan.arena_numel.emplace_back(kv.first, kv.second);
  return plan;
}

}  // namespace

Plan plan_buckets(const std::vector<ParamMeta>& params, const PlanOptions& opt) {
  std::vector<int64_t> order(params.size());
  for (size_t i = 0; i < params.size(); ++i)
    order[i] = opt.reverse ? static_cast<int64_t>(params.size() - 1 - i) : static_cast<int64_t>(i);
  return plan_in_order(params, order, opt);
}

Plan plan_buckets_from_order(const std::vector<ParamMeta>& params,
                             const std::vector<int64_t>& ready_order, const PlanOptions& opt) {
  std::vector<char> seen(params.size(), 0);
  std::vector<int64_t> order;
  order.reserve(params.size());
  for (int64_t idx : ready_order) {
    if (idx < 0 || idx >= static_cast<int64_t>(params.size()))
      throw std::out_of_range("plan_buckets_from_order: parameter index out of range");
    if (seen[static_cast<size_t>(idx)]) continue;
    seen[static_cast<size_t>(idx)] = 1;
    order.push_back(idx);
  }
  for (int64_t i = static_cast<int64_t>(params.size()) - 1; i >= 0; --i)
    if (!seen[static_cast<size_t>(i)]) order.push_back(i);
  return plan_in_order(params, order, opt);
}

std::string describe_plan(const Plan& plan) {
  std::ostringstream os;
  os << plan.buckets.size() << " bucket(s):";
  for (size_t i = 0; i < plan.buckets.size(); ++i) {
    const Bucket& b = plan.buckets[i];
    os << "\n  [" << i << "] group=" << b.group_key << " tensors=" << b.param_indices.size()
       << " numel=" << b.numel << " bytes=" << b.numel * b.elem_size
       << " arena_offset=" << b.arena_offset;
  }
  return os.str();
}

}  // namespace dpa
