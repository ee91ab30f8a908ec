// Fused multi-tensor AdamW for gfx950.
//
// Stands in for torch.optim.AdamW(params, 1e-4) of the reference training loop
// (reference min_DDP.py:74,104), whose stock implementation is a sequence of _foreach_
// kernels.  One launch updates up to AdamWTable::kMaxTensors tensors; when parameters,
// gradients and moments live in flat arenas the table collapses to one entry per bucket.
// The kernel is a pure HBM stream (read p,g,m,v; write p,m,v: 28 B/element in fp32), so it
// is organised for bandwidth only: 16-byte accesses per lane, 4096 elements per workgroup.
//
// Update rule (decoupled weight decay, bias correction, eps outside the square root),
// in the same order as torch:
//   p  = p * (1 - lr*wd)
//   m  = m + (1-b1) * (g - m)
//   v  = b2*v + (1-b2) * g*g
//   p  = p - (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Hyper-parameters and the step counter are read from device memory so that a captured
// hipGraph keeps working when the learning rate changes; `adamw_tick` advances the step
// and refreshes the two bias-correction terms ahead of each update launch.
#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

constexpr int kThreads = 256;

__global__ void adamw_tick_kernel(float* hp) {
  int step = __float_as_int(hp[5]) + 1;
  hp[5] = __int_as_float(step);
  const double b1 = hp[1], b2 = hp[2];
  hp[6] = static_cast<float>(1.0 - pow(b1, static_cast<double>(step)));
  hp[7] = static_cast<float>(sqrt(1.0 - pow(b2, static_cast<double>(step))));
}

__device__ __forceinline__ int find_tensor(const int* block_start, int n, int b) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (block_start[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

struct Hp {
  float lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale;
};

__device__ __forceinline__ void adamw_update(float& p, float g, float& m, float& v, const Hp& h) {
  g *= h.gscale;
  p *= (1.f - h.lr * h.wd);
  m = m + (1.f - h.b1) * (g - m);
  v = h.b2 * v + (1.f - h.b2) * g * g;
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p -= (h.lr / h.bc1) * (m / denom);
}

template <typename P, typename G>
__global__ __launch_bounds__(kThreads) void adamw_kernel(AdamWTable t, const float* __restrict__ hparams,
                                                         const float* __restrict__ grad_scale) {
  const int b = blockIdx.x;
  const int ti = find_tensor(t.block_start, t.n, b);
  const long long base = static_cast<long long>(b - t.block_start[ti]) * AdamWTable::kChunk;
  const long long n = t.numel[ti] - base < AdamWTable::kChunk ? t.numel[ti] - base : AdamWTable::kChunk;

  typename P::T* p = static_cast<typename P::T*>(t.param[ti]) + base;
  const typename G::T* g = static_cast<const typename G::T*>(t.grad[ti]) + base;
  float* m = t.exp_avg[ti] + base;
  float* v = t.exp_avg_sq[ti] + base;
  float* w = t.master[ti] ? t.master[ti] + base : nullptr;

  Hp h;
  h.lr = hparams[0]; h.b1 = hparams[1]; h.b2 = hparams[2]; h.eps = hparams[3]; h.wd = hparams[4];
  h.bc1 = hparams[6]; h.bc2_sqrt = hparams[7];
  h.gscale = grad_scale ? grad_scale[0] : 1.f;

  constexpr int V = 8;
  uintptr_t bits = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                   reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                   reinterpret_cast<uintptr_t>(w);
  if ((bits & 15) == 0) {
    const long long nvec = n / V;
    for (long long i = threadIdx.x; i < nvec; i += kThreads) {
      float pv[V], gv[V], mv[V], vv[V];
      load_as_f32<G, V>(g + i * V, gv);
      load_as_f32<F32, V>(m + i * V, mv);
      load_as_f32<F32, V>(v + i * V, vv);
      if (w) load_as_f32<F32, V>(w + i * V, pv); else load_as_f32<P, V>(p + i * V, pv);
#pragma unroll
      for (int j = 0; j < V; ++j) adamw_update(pv[j], gv[j], mv[j], vv[j], h);
      store_from_f32<F32, V>(m + i * V, mv);
      store_from_f32<F32, V>(v + i * V, vv);
      if (w) store_from_f32<F32, V>(w + i * V, pv);
      store_from_f32<P, V>(p + i * V, pv);
    }
    for (long long i = nvec * V + threadIdx.x; i < n; i += kThreads) {
      float pv = w ? w[i] : P::to_f32(p[i]), mv = m[i], vv = v[i];
      adamw_update(pv, G::to_f32(g[i]), mv, vv, h);
      m[i] = mv; v[i] = vv;
      if (w) w[i] = pv;
      p[i] = P::from_f32(pv);
    }
  } else {
    for (long long i = threadIdx.x; i < n; i += kThreads) {
      float pv = w ? w[i] : P::to_f32(p[i]), mv = m[i], vv = v[i];
      adamw_update(pv, G::to_f32(g[i]), mv, vv, h);
      m[i] = mv; v[i] = vv;
      if (w) w[i] = pv;
      p[i] = P::from_f32(pv);
    }
  }
}

}  // namespace

void launch_adamw_tick(float* hparams, hipStream_t stream) {
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, stream, hparams);
}

void launch_adamw(const AdamWTable& t, int param_dtype, int grad_dtype, const float* hparams,
                  const float* grad_scale, hipStream_t stream) {
  if (t.n <= 0) return;
  const int blocks = t.block_start[t.n];
  if (blocks <= 0) return;
  dispatch_dtype(param_dtype, [&](auto p) {
    dispatch_dtype(grad_dtype, [&](auto g) {
      using P = decltype(p);
      using G = decltype(g);
      hipLaunchKernelGGL((adamw_kernel<P, G>), dim3(blocks), dim3(kThreads), 0, stream, t, hparams, grad_scale);
    });
  });
}

}  // namespace kernels
}  // namespace dpa
