// Fused softmax cross-entropy with argmax-accuracy for gfx950.
//
// Replaces, in one pass over the logits, what the reference does with
// nn.CrossEntropyLoss (log_softmax + nll_loss, reference min_DDP.py:75,100) plus
// torch.argmax(y_hat, 1).eq(y) (reference min_DDP.py:106): each row yields its loss, its
// log-sum-exp (kept for the backward) and a "prediction correct" flag.  The backward is
// a second streaming pass writing dlogits = (softmax - onehot) * upstream, optionally over
// the logits buffer itself.
//
// Two shapes of row:
//   * small C (<= 64, the reference's n_classes = 4): one lane per row, the whole row in
//     registers, 64 rows per wave -- no cross-lane traffic at all.
//   * large C (ImageNet 1000, GPT-2 50257 padded to 50304): one workgroup per row,
//     online max / sum-of-exponentials over 16-byte vectors (8 bf16 per lane per access),
//     one rescale exponential per vector rather than per element, wave shuffles + LDS for
//     the final combine.  The pass is HBM-bound: 2 B/element in, nothing out but 9 B/row.
#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {
#ifndef XENT_STREAM
#define XENT_STREAM 1
#endif
constexpr bool kXentStream = XENT_STREAM != 0;


constexpr float kNegInf = -__builtin_huge_valf();

// ---------------------------------------------------------------- small C: lane per row
template <typename TR>
__global__ __launch_bounds__(256) void xent_fwd_small_kernel(const typename TR::T* __restrict__ logits,
                                                             const long long* __restrict__ targets, long long N,
                                                             int C, long long ld, long long ignore_index,
                                                             float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                             unsigned char* __restrict__ correct) {
  const long long r = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const typename TR::T* row = logits + r * ld;
  const long long tgt = targets[r];
  float mx = kNegInf, xt = 0.f;
  int arg = 0;
  for (int j = 0; j < C; ++j) {
    const float x = TR::to_f32(row[j]);
    if (x > mx) { mx = x; arg = j; }
    if (j == tgt) xt = x;
  }
  float s = 0.f;
  for (int j = 0; j < C; ++j) s += __expf(TR::to_f32(row[j]) - mx);
  // max and log-sum are kept apart: (max - x_t) + log(sum) stays accurate when |logits| is
  // large, where max + log(sum) would round at the magnitude of the logits.
  const float ls = __logf(s);
  const bool ignored = (tgt == ignore_index) || tgt < 0 || tgt >= C;
  row_lse[2 * r] = mx;
  row_lse[2 * r + 1] = ls;
  row_loss[r] = ignored ? 0.f : (mx - xt) + ls;
  correct[r] = (!ignored && arg == tgt) ? 1 : 0;
}

template <typename TR>
__global__ __launch_bounds__(256) void xent_bwd_small_kernel(const typename TR::T* logits, const long long* __restrict__ targets,
                                                             const float* __restrict__ row_lse, const float* __restrict__ grad_out,
                                                             const float* __restrict__ metrics, int mean, long long N, int C,
                                                             long long ld, long long ignore_index, typename TR::T* dlogits) {
  const long long r = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const long long tgt = targets[r];
  const bool ignored = (tgt == ignore_index) || tgt < 0 || tgt >= C;
  float g = grad_out[0];
  if (mean) g /= fmaxf(metrics[2], 1.f);
  if (ignored) g = 0.f;
  const float mx = row_lse[2 * r], ls = row_lse[2 * r + 1];
  const typename TR::T* row = logits + r * ld;
  typename TR::T* out = dlogits + r * ld;
  for (int j = 0; j < C; ++j) {
    const float p = __expf((TR::to_f32(row[j]) - mx) - ls);
    out[j] = TR::from_f32((p - (j == tgt ? 1.f : 0.f)) * g);
  }
  for (long long j = C; j < ld; ++j) out[j] = TR::from_f32(0.f);
}

// ---------------------------------------------------------------- large C: workgroup per row
// FUSE_BWD (training): the same workgroup then walks its row a second time and writes
// (softmax - onehot) * grad_scale[0] to `dlogits` (which may be the logits buffer itself).  The row
// was read a moment ago, and with 1024-thread workgroups only ~512 rows of 100 KB are in flight on
// the chip, so the second read is served by the L2 / Infinity Cache: the [tokens, vocabulary] tensor
// crosses HBM twice (read, write) instead of three times as with separate forward and backward passes.
template <typename TR, int THREADS, bool FUSE_BWD = false>
__global__ __launch_bounds__(THREADS) void xent_fwd_row_kernel(const typename TR::T* logits,
                                                               const long long* __restrict__ targets, long long C,
                                                               long long ld, long long ignore_index,
                                                               float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                               unsigned char* __restrict__ correct,
                                                               const float* __restrict__ grad_scale = nullptr,
                                                               typename TR::T* dlogits = nullptr) {
  constexpr int V = TR::kVec;
  constexpr int NW = THREADS / 64;
  __shared__ float s_m[NW], s_s[NW];
  __shared__ long long s_a[NW];
  __shared__ float s_stats[2];
  const long long r = blockIdx.x;
  const typename TR::T* row = logits + r * ld;
  const long long tgt = targets[r];

  float m = kNegInf, s = 0.f;
  long long arg = 0x7fffffffffffffffll;
  const bool aligned = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
  const long long nvec = aligned ? C / V : 0;
  // Two vectors in flight per lane per trip to keep HBM requests outstanding.
  long long i = threadIdx.x;
  for (; i + THREADS < nvec; i += 2 * THREADS) {
    float a[V], b[V];
    load_as_f32<TR, V>(row + i * V, a);
    load_as_f32<TR, V>(row + (i + THREADS) * V, b);
    float vm = a[0];
    int va = 0;
#pragma unroll
    for (int j = 1; j < V; ++j) if (a[j] > vm) { vm = a[j]; va = j; }
    float wm = b[0];
    int wa = 0;
#pragma unroll
    for (int j = 1; j < V; ++j) if (b[j] > wm) { wm = b[j]; wa = j; }
    long long cand = i * V + va;
    if (wm > vm) { vm = wm; cand = (i + THREADS) * V + wa; }
    if (vm > m) {
      s *= __expf(m - vm);  // exp(-inf) = 0 on the first trip
      m = vm;
      arg = cand;
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) acc += __expf(a[j] - m) + __expf(b[j] - m);
    s += acc;
  }
  for (; i < nvec; i += THREADS) {
    float a[V];
    load_as_f32<TR, V>(row + i * V, a);
    float vm = a[0];
    int va = 0;
#pragma unroll
    for (int j = 1; j < V; ++j) if (a[j] > vm) { vm = a[j]; va = j; }
    if (vm > m) {
      s *= __expf(m - vm);
      m = vm;
      arg = i * V + va;
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) acc += __expf(a[j] - m);
    s += acc;
  }
  for (long long j = nvec * V + threadIdx.x; j < C; j += THREADS) {
    const float x = TR::to_f32(row[j]);
    if (x > m) {
      s *= __expf(m - x);
      m = x;
      arg = j;
    }
    s += __expf(x - m);
  }

  // Combine (m, s, arg) across the wave, then across waves.  Ties go to the lower index.
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(m, off, 64);
    const float os = __shfl_xor(s, off, 64);
    const long long oa = __shfl_xor(arg, off, 64);
    const float nm = fmaxf(m, om);
    const float sa = (m == kNegInf) ? 0.f : s * __expf(m - nm);
    const float sb = (om == kNegInf) ? 0.f : os * __expf(om - nm);
    if (om > m || (om == m && oa < arg)) arg = oa;
    m = nm;
    s = sa + sb;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { s_m[wid] = m; s_s[wid] = s; s_a[wid] = arg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = s_m[0], S = s_s[0];
    long long A = s_a[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float om = s_m[w], os = s_s[w];
      const long long oa = s_a[w];
      const float nm = fmaxf(M, om);
      const float sa = (M == kNegInf) ? 0.f : S * __expf(M - nm);
      const float sb = (om == kNegInf) ? 0.f : os * __expf(om - nm);
      if (om > M || (om == M && oa < A)) A = oa;
      M = nm;
      S = sa + sb;
    }
    const float ls = __logf(S);
    const bool ignored = (tgt == ignore_index) || tgt < 0 || tgt >= C;
    if (row_lse) {
      row_lse[2 * r] = M;
      row_lse[2 * r + 1] = ls;
    }
    row_loss[r] = ignored ? 0.f : (M - TR::to_f32(row[tgt])) + ls;
    correct[r] = (!ignored && A == tgt) ? 1 : 0;
    if (FUSE_BWD) {
      s_stats[0] = M;
      s_stats[1] = ls;
    }
  }
  if (FUSE_BWD) {
    __syncthreads();  // statistics are published, and row[tgt] was read before the row is overwritten
    const float mx = s_stats[0], ls = s_stats[1];
    const bool ignored = (tgt == ignore_index) || tgt < 0 || tgt >= C;
    const float g = ignored ? 0.f : grad_scale[0];
    typename TR::T* out = dlogits + r * ld;
    const bool aligned2 = ((reinterpret_cast<uintptr_t>(row) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const long long nvec2 = aligned2 ? C / V : 0;
    for (long long k = threadIdx.x; k < nvec2; k += THREADS) {
      float a[V];
      load_as_f32<TR, V>(row + k * V, a);
      const long long j0 = k * V;
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] = (__expf((a[j] - mx) - ls) - ((j0 + j) == tgt ? 1.f : 0.f)) * g;
      store_from_f32<TR, V>(out + k * V, a);
    }
    for (long long j = nvec2 * V + threadIdx.x; j < ld; j += THREADS) {
      float d = 0.f;
      if (j < C) d = (__expf((TR::to_f32(row[j]) - mx) - ls) - (j == tgt ? 1.f : 0.f)) * g;
      out[j] = TR::from_f32(d);
    }
  }
}

// The fused kernel above for rows that fit in the workgroup's registers (16-bit logits, at most KEEP
// 16-byte vectors per thread: a GPT-2 vocabulary row is 6288 vectors, 6.1 per thread of 1024): the row is
// loaded ONCE, kept packed (4 registers per vector), reduced, turned into its gradient and stored, so the
// tensor crosses the memory system exactly twice and nothing depends on the second read hitting a cache.
// Requires 16-byte aligned rows of ld % 8 == 0 elements; columns C..ld-1 are padding (left out of the
// softmax, zero-filled on output).
template <typename TR, int THREADS, int KEEP>
__global__ __launch_bounds__(THREADS, THREADS / 128) void xent_fused_resident_kernel(typename TR::T* logits, const long long* __restrict__ targets,
                                                                      long long C, long long ld, long long ignore_index,
                                                                      float* __restrict__ row_loss, unsigned char* __restrict__ correct,
                                                                      const float* __restrict__ grad_scale) {
  static_assert(TR::kVec == 8, "16-bit element types only");
  constexpr int NW = THREADS / 64;
  __shared__ float s_m[NW], s_s[NW], s_t[NW];
  __shared__ int s_a[NW];
  __shared__ float s_stats[2];
  const long long r = blockIdx.x;
  typename TR::T* row = logits + r * ld;
  const long long tgt = targets[r];
  const bool ignored = (tgt == ignore_index) || tgt < 0 || tgt >= C;
  const int nvec = static_cast<int>((C + 7) / 8), nvec_ld = static_cast<int>(ld / 8);
  const int cfull = static_cast<int>(C / 8), crem = static_cast<int>(C % 8);  // vector cfull holds crem real columns
  const int tvec = ignored ? -1 : static_cast<int>(tgt >> 3), tsub = static_cast<int>(tgt & 7);

  char* row_bytes = reinterpret_cast<char*>(row);
  raw16 kept[KEEP];
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int k = threadIdx.x + q * THREADS;
    if (k < nvec) {  // uniform base + 32-bit offset; read once
      if (kXentStream) kept[q] = __builtin_nontemporal_load(reinterpret_cast<const raw16*>(row_bytes + 16u * static_cast<unsigned>(k)));
      else kept[q] = *reinterpret_cast<const raw16*>(row_bytes + 16u * static_cast<unsigned>(k));
    }
  }
  auto element = [](const raw16& v, int j) { return TR::to_f32(static_cast<u16>(v[j >> 1] >> (16 * (j & 1)))); };

  // this thread's maximum (lowest index among equals) and the target's logit if it holds it
  float m = kNegInf, xt = 0.f;
  int arg = 0x7fffffff;
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int k = threadIdx.x + q * THREADS;
    if (k < nvec) {
      const int nval = k < cfull ? 8 : crem;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = j < nval ? element(kept[q], j) : kNegInf;
        if (x > m) { m = x; arg = 8 * k + j; }
        if (k == tvec && j == tsub) xt = x;
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // one vector at a time: interleaved, the expansions of all KEEP vectors would be live at once
  }
  // (the packed row is made opaque between the passes: otherwise the compiler keeps the fp32 expansions of
  // the first pass for the later ones and spills them)
  auto forget_expansions = [&]() {
#pragma unroll
    for (int q = 0; q < KEEP; ++q)
#pragma unroll
      for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(kept[q][w]));
  };
  forget_expansions();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int k = threadIdx.x + q * THREADS;
    if (k < nvec) {
      const int nval = k < cfull ? 8 : crem;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += j < nval ? __expf(element(kept[q], j) - m) : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // Combine (m, s, arg, target logit) across the wave, then across waves.  Ties go to the lower index.
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(m, off, 64);
    const float os = __shfl_xor(s, off, 64);
    const int oa = __shfl_xor(arg, off, 64);
    xt += __shfl_xor(xt, off, 64);
    const float nm = fmaxf(m, om);
    const float sa = (m == kNegInf) ? 0.f : s * __expf(m - nm);
    const float sb = (om == kNegInf) ? 0.f : os * __expf(om - nm);
    if (om > m || (om == m && oa < arg)) arg = oa;
    m = nm;
    s = sa + sb;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { s_m[wid] = m; s_s[wid] = s; s_a[wid] = arg; s_t[wid] = xt; }
  __syncthreads();
  if (wid == 0) {  // the first NW lanes combine the waves' results with the same exchange
    static_assert(NW <= 64 && (NW & (NW - 1)) == 0, "one lane per wave, power of two");
    const bool has = lane < NW;
    m = has ? s_m[lane] : kNegInf;
    s = has ? s_s[lane] : 0.f;
    arg = has ? s_a[lane] : 0x7fffffff;
    xt = has ? s_t[lane] : 0.f;
#pragma unroll
    for (int off = NW / 2; off > 0; off >>= 1) {
      const float om = __shfl_xor(m, off, 64);
      const float os = __shfl_xor(s, off, 64);
      const int oa = __shfl_xor(arg, off, 64);
      xt += __shfl_xor(xt, off, 64);
      const float nm = fmaxf(m, om);
      const float sa = (m == kNegInf) ? 0.f : s * __expf(m - nm);
      const float sb = (om == kNegInf) ? 0.f : os * __expf(om - nm);
      if (om > m || (om == m && oa < arg)) arg = oa;
      m = nm;
      s = sa + sb;
    }
    if (lane == 0) {
      const float ls = __logf(s);
      row_loss[r] = ignored ? 0.f : (m - xt) + ls;
      correct[r] = (!ignored && arg == tgt) ? 1 : 0;
      s_stats[0] = m;
      s_stats[1] = ls;
    }
  }
  __syncthreads();
  forget_expansions();
  const float shift = s_stats[0] + s_stats[1];  // x - max - log(sum)
  const float g = ignored ? 0.f : grad_scale[0];
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int k = threadIdx.x + q * THREADS;
    if (k < nvec) {
      const int nval = k < cfull ? 8 : crem;
      float d[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        d[j] = j < nval ? (__expf(element(kept[q], j) - shift) - ((k == tvec && j == tsub) ? 1.f : 0.f)) * g : 0.f;
      store_from_f32<TR, 8, kXentStream>(reinterpret_cast<typename TR::T*>(row_bytes + 16u * static_cast<unsigned>(k)), d);
    } else if (k < nvec_ld) {
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      store_from_f32<TR, 8, kXentStream>(reinterpret_cast<typename TR::T*>(row_bytes + 16u * static_cast<unsigned>(k)), z);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// grad_scale[0] = 1 / max(#targets != ignore_index, 1) for a mean, 1 for a sum.
__global__ __launch_bounds__(1024) void xent_scale_kernel(const long long* __restrict__ targets, long long N, long long ignore_index,
                                                          int mean, float* __restrict__ grad_scale) {
  __shared__ float scratch[16];
  // One workgroup, fixed summation order; every thread takes four consecutive targets per step (two
  // 16-byte loads) and the steps are independent, so the loop is bound by bandwidth, not by load latency.
  float valid = 0.f;
  const long long n4 = (reinterpret_cast<uintptr_t>(targets) & 15) == 0 ? N / 4 : 0;
#pragma unroll 4
  for (long long i = threadIdx.x; i < n4; i += 1024) {
    const longlong2 a = *reinterpret_cast<const longlong2*>(targets + 4 * i);
    const longlong2 b = *reinterpret_cast<const longlong2*>(targets + 4 * i + 2);
    valid += (a.x != ignore_index ? 1.f : 0.f) + (a.y != ignore_index ? 1.f : 0.f) + (b.x != ignore_index ? 1.f : 0.f) +
             (b.y != ignore_index ? 1.f : 0.f);
  }
  for (long long i = 4 * n4 + threadIdx.x; i < N; i += 1024) valid += (targets[i] != ignore_index) ? 1.f : 0.f;
  valid = block_sum<16>(valid, scratch);
  if (threadIdx.x == 0) grad_scale[0] = mean ? 1.f / fmaxf(valid, 1.f) : 1.f;
}

template <typename TR, int THREADS>
__global__ __launch_bounds__(THREADS) void xent_bwd_row_kernel(const typename TR::T* logits, const long long* __restrict__ targets,
                                                               const float* __restrict__ row_lse, const float* __restrict__ grad_out,
                                                               const float* __restrict__ metrics, int mean, long long C,
                                                               long long ld, long long ignore_index, typename TR::T* dlogits) {
  constexpr int V = TR::kVec;
  const long long r = blockIdx.x;
  const long long tgt = targets[r];
  const bool ignored = (tgt == ignore_index) || tgt < 0 || tgt >= C;
  float g = grad_out[0];
  if (mean) g /= fmaxf(metrics[2], 1.f);
  if (ignored) g = 0.f;
  const float mx = row_lse[2 * r], ls = row_lse[2 * r + 1];
  const typename TR::T* row = logits + r * ld;
  typename TR::T* out = dlogits + r * ld;
  const bool aligned = ((reinterpret_cast<uintptr_t>(row) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const long long nvec = aligned ? C / V : 0;
  for (long long i = threadIdx.x; i < nvec; i += THREADS) {
    float a[V];
    load_as_f32<TR, V>(row + i * V, a);
    const long long j0 = i * V;
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = (__expf((a[j] - mx) - ls) - ((j0 + j) == tgt ? 1.f : 0.f)) * g;
    store_from_f32<TR, V>(out + i * V, a);
  }
  for (long long j = nvec * V + threadIdx.x; j < ld; j += THREADS) {
    float d = 0.f;
    if (j < C) d = (__expf((TR::to_f32(row[j]) - mx) - ls) - (j == tgt ? 1.f : 0.f)) * g;
    out[j] = TR::from_f32(d);
  }
}

// ---------------------------------------------------------------- metric reduction
__global__ __launch_bounds__(1024) void xent_reduce_kernel(const float* __restrict__ row_loss,
                                                           const unsigned char* __restrict__ correct,
                                                           const long long* __restrict__ targets, long long N,
                                                           long long ignore_index, float* __restrict__ metrics) {
  __shared__ float scratch[16];
  // One workgroup, fixed summation order (bit-reproducible metrics); four consecutive rows per thread and
  // step with 16-byte / 4-byte vector loads, independent steps.
  float loss = 0.f, corr = 0.f, valid = 0.f;
  const bool aligned = ((reinterpret_cast<uintptr_t>(row_loss) | reinterpret_cast<uintptr_t>(targets)) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(correct) & 3) == 0;
  const long long n4 = aligned ? N / 4 : 0;
#pragma unroll 4
  for (long long i = threadIdx.x; i < n4; i += 1024) {
    const f32x4 l = *reinterpret_cast<const f32x4*>(row_loss + 4 * i);
    const unsigned c = *reinterpret_cast<const unsigned*>(correct + 4 * i);
    const longlong2 a = *reinterpret_cast<const longlong2*>(targets + 4 * i);
    const longlong2 b = *reinterpret_cast<const longlong2*>(targets + 4 * i + 2);
    loss += (l[0] + l[1]) + (l[2] + l[3]);
    corr += static_cast<float>(__builtin_popcount(c & 0x01010101u));
    valid += (a.x != ignore_index ? 1.f : 0.f) + (a.y != ignore_index ? 1.f : 0.f) + (b.x != ignore_index ? 1.f : 0.f) +
             (b.y != ignore_index ? 1.f : 0.f);
  }
  for (long long i = 4 * n4 + threadIdx.x; i < N; i += 1024) {
    loss += row_loss[i];
    corr += correct[i] ? 1.f : 0.f;
    valid += (targets[i] != ignore_index) ? 1.f : 0.f;
  }
  loss = block_sum<16>(loss, scratch);
  corr = block_sum<16>(corr, scratch);
  valid = block_sum<16>(valid, scratch);
  if (threadIdx.x == 0) {
    metrics[0] = loss;
    metrics[1] = corr;
    metrics[2] = valid;
    metrics[3] = loss / fmaxf(valid, 1.f);
  }
}

constexpr int kSmallC = 64;

}  // namespace

void launch_xent_fwd(const void* logits, int dtype, const long long* targets, long long N, long long C, long long ld,
                     long long ignore_index, float* row_loss, float* row_lse, unsigned char* correct, hipStream_t stream) {
  if (N <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    const typename TR::T* lg = static_cast<const typename TR::T*>(logits);
    if (C <= kSmallC) {
      const int blocks = static_cast<int>((N + 255) / 256);
      hipLaunchKernelGGL((xent_fwd_small_kernel<TR>), dim3(blocks), dim3(256), 0, stream, lg, targets, N,
                         static_cast<int>(C), ld, ignore_index, row_loss, row_lse, correct);
    } else if (C <= 2048) {
      hipLaunchKernelGGL((xent_fwd_row_kernel<TR, 64>), dim3(static_cast<unsigned>(N)), dim3(64), 0, stream, lg, targets, C,
                         ld, ignore_index, row_loss, row_lse, correct);
    } else {
      hipLaunchKernelGGL((xent_fwd_row_kernel<TR, 256>), dim3(static_cast<unsigned>(N)), dim3(256), 0, stream, lg, targets,
                         C, ld, ignore_index, row_loss, row_lse, correct);
    }
  });
}

void launch_xent_reduce(const float* row_loss, const unsigned char* correct, const long long* targets, long long N,
                        long long ignore_index, float* metrics, hipStream_t stream) {
  hipLaunchKernelGGL(xent_reduce_kernel, dim3(1), dim3(1024), 0, stream, row_loss, correct, targets, N, ignore_index,
                     metrics);
}

void launch_xent_bwd(const void* logits, int dtype, const long long* targets, const float* row_lse, const float* grad_out,
                     const float* metrics, int mean, long long N, long long C, long long ld, long long ignore_index,
                     void* dlogits, hipStream_t stream) {
  if (N <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    const typename TR::T* lg = static_cast<const typename TR::T*>(logits);
    typename TR::T* dl = static_cast<typename TR::T*>(dlogits);
    if (C <= kSmallC) {
      const int blocks = static_cast<int>((N + 255) / 256);
      hipLaunchKernelGGL((xent_bwd_small_kernel<TR>), dim3(blocks), dim3(256), 0, stream, lg, targets, row_lse, grad_out,
                         metrics, mean, N, static_cast<int>(C), ld, ignore_index, dl);
    } else if (C <= 2048) {
      hipLaunchKernelGGL((xent_bwd_row_kernel<TR, 64>), dim3(static_cast<unsigned>(N)), dim3(64), 0, stream, lg, targets,
                         row_lse, grad_out, metrics, mean, C, ld, ignore_index, dl);
    } else {
      hipLaunchKernelGGL((xent_bwd_row_kernel<TR, 256>), dim3(static_cast<unsigned>(N)), dim3(256), 0, stream, lg, targets,
                         row_lse, grad_out, metrics, mean, C, ld, ignore_index, dl);
    }
  });
}

bool xent_fused_supported(long long C) { return C > kSmallC; }

void launch_xent_fused(void* logits, int dtype, const long long* targets, long long N, long long C, long long ld,
                       long long ignore_index, int mean, float* grad_scale, float* row_loss, unsigned char* correct,
                       hipStream_t stream) {
  if (N <= 0) return;
  hipLaunchKernelGGL(xent_scale_kernel, dim3(1), dim3(1024), 0, stream, targets, N, ignore_index, mean, grad_scale);
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    T* lg = static_cast<T*>(logits);
    constexpr int kKeep = 7;
    if constexpr (TR::kVec == 8) {
      if (C > 4096 && ld % 8 == 0 && ld / 8 <= kKeep * 1024 && (reinterpret_cast<uintptr_t>(lg) & 15) == 0) {
        hipLaunchKernelGGL((xent_fused_resident_kernel<TR, 1024, kKeep>), dim3(static_cast<unsigned>(N)), dim3(1024), 0, stream, lg, targets,
                           C, ld, ignore_index, row_loss, correct, grad_scale);
        return;
      }
    }
    if (C <= 4096) {
      hipLaunchKernelGGL((xent_fwd_row_kernel<TR, 256, true>), dim3(static_cast<unsigned>(N)), dim3(256), 0, stream, lg, targets, C, ld,
                         ignore_index, row_loss, static_cast<float*>(nullptr), correct, grad_scale, lg);
    } else {
      hipLaunchKernelGGL((xent_fwd_row_kernel<TR, 1024, true>), dim3(static_cast<unsigned>(N)), dim3(1024), 0, stream, lg, targets, C, ld,
                         ignore_index, row_loss, static_cast<float*>(nullptr), correct, grad_scale, lg);
    }
  });
}

}  // namespace kernels
}  // namespace dpa
