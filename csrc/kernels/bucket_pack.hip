// Gradient bucket flatten / unflatten for gfx950.
//
// One launch moves up to CopyTable::kMaxTensors tensors between their own storage and a
// flat bucket: dst = cast(src * scale).  This is the copy the c10d reducer performs per
// parameter with `bucket.copy_(grad); bucket.div_(world)` when the reference wraps its model
// in DistributedDataParallel (reference distributed.py:112-115); here it is a single
// HBM-bound pass: 16 bytes per lane per access, 8192 elements per workgroup so a 64 MiB
// bucket is ~2048 workgroups (8 per CU), and the pre-divide and the fp32<->bf16 cast ride along.
#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int find_tensor(const int* block_start, int n, int b) {
  int lo = 0, hi = n;  // invariant: block_start[lo] <= b < block_start[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (block_start[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename S, typename D>
__global__ __launch_bounds__(kThreads) void multi_copy_scale_kernel(CopyTable t, float scale) {
  const int b = blockIdx.x;
  const int ti = find_tensor(t.block_start, t.n, b);
  const long long base = static_cast<long long>(b - t.block_start[ti]) * CopyTable::kChunk;
  const long long n = t.numel[ti] - base < CopyTable::kChunk ? t.numel[ti] - base : CopyTable::kChunk;
  const typename S::T* src = static_cast<const typename S::T*>(t.src[ti]) + base;
  typename D::T* dst = static_cast<typename D::T*>(t.dst[ti]) + base;

  constexpr int V = 8;  // elements per lane per step: 16 B of bf16, 32 B of fp32
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (aligned) {
    const long long nvec = n / V;
    for (long long i = threadIdx.x; i < nvec; i += kThreads) {
      float v[V];
      load_as_f32<S, V>(src + i * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] *= scale;
      store_from_f32<D, V>(dst + i * V, v);
    }
    for (long long i = nvec * V + threadIdx.x; i < n; i += kThreads)
      dst[i] = D::from_f32(S::to_f32(src[i]) * scale);
  } else {
    for (long long i = threadIdx.x; i < n; i += kThreads)
      dst[i] = D::from_f32(S::to_f32(src[i]) * scale);
  }
}

template <typename S>
__global__ __launch_bounds__(kThreads) void scale_inplace_kernel(typename S::T* p, long long n, float scale) {
  constexpr int V = 8;
  const long long nvec = n / V;
  const long long stride = static_cast<long long>(gridDim.x) * kThreads;
  const long long tid = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    for (long long i = tid; i < nvec; i += stride) {
      float v[V];
      load_as_f32<S, V>(p + i * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] *= scale;
      store_from_f32<S, V>(p + i * V, v);
    }
    for (long long i = nvec * V + tid; i < n; i += stride) p[i] = S::from_f32(S::to_f32(p[i]) * scale);
  } else {
    for (long long i = tid; i < n; i += stride) p[i] = S::from_f32(S::to_f32(p[i]) * scale);
  }
}

template <typename S>
__global__ __launch_bounds__(kThreads) void multi_sumsq_kernel(CopyTable t, float* partials) {
  __shared__ float scratch[kThreads / 64];
  const int b = blockIdx.x;
  const int ti = find_tensor(t.block_start, t.n, b);
  const long long base = static_cast<long long>(b - t.block_start[ti]) * CopyTable::kChunk;
  const long long n = t.numel[ti] - base < CopyTable::kChunk ? t.numel[ti] - base : CopyTable::kChunk;
  const typename S::T* src = static_cast<const typename S::T*>(t.src[ti]) + base;
  constexpr int V = 8;
  float acc = 0.f;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const long long nvec = n / V;
    for (long long i = threadIdx.x; i < nvec; i += kThreads) {
      float v[V];
      load_as_f32<S, V>(src + i * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) acc += v[j] * v[j];
    }
    for (long long i = nvec * V + threadIdx.x; i < n; i += kThreads) {
      float x = S::to_f32(src[i]);
      acc += x * x;
    }
  } else {
    for (long long i = threadIdx.x; i < n; i += kThreads) {
      float x = S::to_f32(src[i]);
      acc += x * x;
    }
  }
  acc = block_sum<kThreads / 64>(acc, scratch);
  // One partial per workgroup; the final sum walks them in a fixed order (no float atomics, so the
  // gradient norm is bit-identical from run to run).
  if (threadIdx.x == 0) partials[b] = acc;
}

__global__ __launch_bounds__(kThreads) void sum_partials_kernel(const float* partials, int n, float* out) {
  __shared__ float scratch[kThreads / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += kThreads) acc += partials[i];
  acc = block_sum<kThreads / 64>(acc, scratch);
  if (threadIdx.x == 0) out[0] = acc;
}

__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* coef, float* norm_out) {
  const float norm = sqrtf(sumsq[0]);
  if (norm_out) norm_out[0] = norm;
  const float c = max_norm / (norm + 1e-6f);
  coef[0] = c < 1.f ? c : 1.f;
}

}  // namespace

void launch_multi_copy_scale(const CopyTable& t, int src_dtype, int dst_dtype, float scale, hipStream_t stream) {
  if (t.n <= 0) return;
  const int blocks = t.block_start[t.n];
  if (blocks <= 0) return;
  dispatch_dtype(src_dtype, [&](auto s) {
    dispatch_dtype(dst_dtype, [&](auto d) {
      using S = decltype(s);
      using D = decltype(d);
      hipLaunchKernelGGL((multi_copy_scale_kernel<S, D>), dim3(blocks), dim3(kThreads), 0, stream, t, scale);
    });
  });
}

void launch_scale_inplace(void* p, long long n, int dtype, float scale, hipStream_t stream) {
  if (n <= 0) return;
  long long want = (n / 8 + kThreads - 1) / kThreads;
  int blocks = static_cast<int>(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  dispatch_dtype(dtype, [&](auto s) {
    using S = decltype(s);
    hipLaunchKernelGGL((scale_inplace_kernel<S>), dim3(blocks), dim3(kThreads), 0, stream,
                       static_cast<typename S::T*>(p), n, scale);
  });
}

void launch_sumsq(const CopyTable& t, int dtype, float* partials, hipStream_t stream) {
  if (t.n <= 0) return;
  const int blocks = t.block_start[t.n];
  if (blocks <= 0) return;
  dispatch_dtype(dtype, [&](auto s) {
    using S = decltype(s);
    hipLaunchKernelGGL((multi_sumsq_kernel<S>), dim3(blocks), dim3(kThreads), 0, stream, t, partials);
  });
}

void launch_sum_partials(const float* partials, int n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kThreads), 0, stream, partials, n, out);
}

void launch_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, hipStream_t stream) {
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, max_norm, coef, norm_out);
}

}  // namespace kernels
}  // namespace dpa
