// Gradient bucket planner.
//
// Plays the role of torch's compute_bucket_assignment_by_size, which the reference
// reaches through DistributedDataParallel(...) (reference distributed.py:112-115), but
// the output is a *memory layout*, not just an index partition: each bucket is one
// contiguous slab of a flat arena, every tensor starts on an `align_bytes` boundary so
// the pack / optimizer kernels can use 16-byte vector accesses per lane, and bucket
// sizes default to values chosen for RCCL over a fully connected 8-GPU xGMI mesh
// (per-link payload is bucket/8, so buckets are larger than the 25 MiB NVLink default).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace dpa {

struct ParamMeta {
  int64_t numel = 0;
  int32_t elem_size = 4;   // bytes per element of the *bucket* dtype for this tensor
  int32_t group_key = 0;   // tensors only share a bucket when their key matches (dtype/device)
  bool requires_grad = true;
};

struct Bucket {
  int32_t group_key = 0;
  int32_t elem_size = 4;
  std::vector<int64_t> param_indices;  // indices into the caller's parameter list
  std::vector<int64_t> offsets;        // element offset of each tensor inside the bucket
  int64_t numel = 0;                   // padded element count of the bucket
  int64_t arena_offset = 0;            // element offset of the bucket inside its group's arena
};

struct PlanOptions {
  int64_t first_bucket_bytes = 4ll << 20;
  int64_t bucket_bytes = 64ll << 20;
  int64_t align_bytes = 128;
  bool reverse = true;  // walk parameters last-to-first: the order backward produces grads
};

struct Plan {
  std::vector<Bucket> buckets;
  // Total padded elements per group key (size of each flat arena).
  std::vector<std::pair<int32_t, int64_t>> arena_numel;
};

Plan plan_buckets(const std::vector<ParamMeta>& params, const PlanOptions& opt);

// Re-plan from an observed gradient-ready order (first backward pass), keeping the
// same size policy.  `ready_order` lists parameter indices in the order their
// gradients became ready; parameters missing from it are appended in reverse index order.
Plan plan_buckets_from_order(const std::vector<ParamMeta>& params,
                             const std::vector<int64_t>& ready_order, const PlanOptions& opt);

std::string describe_plan(const Plan& plan);

}  // namespace dpa
