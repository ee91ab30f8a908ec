// The reference model as ONE kernel per direction, for gfx950.
//
// The reference network is Linear(1, H) -> Linear(H, C) with C = 4 and nothing in between
// (reference min_DDP.py:41-49).  Run layer by layer, its [M, H] hidden activation (1 GB at
// M = 65536, H = 4096 in fp32) crosses HBM five times per training step although every element
// of it is a single FMA of one input scalar: h[m][k] = x[m] * w1[k] + b1[k].  Here the hidden
// layer is never stored:
//   forward   y[m][n]  = b2[n] + sum_k act(x[m] w1[k] + b1[k]) * W2[n][k]
//             a lane owns a row (x[m] and the C running sums live in registers), the four waves
//             of a workgroup split k, and all weight operands are wave-uniform, so they arrive
//             through the scalar data path and every vector instruction is an FMA;
//   backward  dW2[n][k] = sum_m dy[m][n] h[m][k]          dh[m][k] = act'(.) * sum_n dy[m][n] W2[n][k]
//             dw1[k]    = sum_m dh[m][k] x[m]              db1[k]    = sum_m dh[m][k]
//             db2[n]    = sum_m dy[m][n]
//             a lane owns a hidden unit k (its w1, b1 and W2 column in registers), a workgroup
//             walks a chunk of rows whose x[m] / dy[m][:] are wave-uniform, recomputes h and dh
//             on the fly, and writes per-chunk partial sums; a second pass adds the chunks up in a
//             fixed order (no float atomics: results are bit-reproducible).
// HBM traffic per step drops from ~5 GB to the [M, C] logits and their gradient (a few MB).
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

template <int ACT>
__device__ __forceinline__ float hidden_act(float h) {
  if constexpr (ACT == kActRelu) return h > 0.f ? h : 0.f;
  return h;
}

// grid.x = row blocks of 64 rows; wave w of the workgroup covers k in [w * kq, (w + 1) * kq).
template <typename TR, int NT, int ACT>
__global__ __launch_bounds__(kThreads) void rank1_mlp_fwd_kernel(const typename TR::T* __restrict__ x,
                                                                 const typename TR::T* __restrict__ w1,
                                                                 const typename TR::T* __restrict__ b1,
                                                                 const typename TR::T* __restrict__ W2,
                                                                 const typename TR::T* __restrict__ b2,
                                                                 typename TR::T* __restrict__ y, long long M, int H, int N) {
  __shared__ float red[kWaves][NT][64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long m = static_cast<long long>(blockIdx.x) * 64 + lane;
  const float xv = m < M ? TR::to_f32(x[m]) : 0.f;
  const int kq = (H + kWaves - 1) / kWaves;
  const int k0 = wid * kq, k1 = (k0 + kq < H) ? k0 + kq : H;
  float acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = 0.f;
  // rows of W2 beyond N alias the last real row: their sums are computed and thrown away
  const typename TR::T* w2row[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) w2row[n] = W2 + static_cast<long long>(n < N ? n : N - 1) * H;
#pragma unroll 8
  for (int k = k0; k < k1; ++k) {
    const float h = hidden_act<ACT>(fmaf(xv, TR::to_f32(w1[k]), b1 ? TR::to_f32(b1[k]) : 0.f));
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = fmaf(h, TR::to_f32(w2row[n][k]), acc[n]);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) red[wid][n][lane] = acc[n];
  __syncthreads();
  // 64 rows x NT outputs, folded over the four k-quarters in a fixed order
  for (int i = threadIdx.x; i < 64 * NT; i += kThreads) {
    const int r = i / NT, n = i % NT;
    const long long mr = static_cast<long long>(blockIdx.x) * 64 + r;
    if (n < N && mr < M) {
      float s = b2 ? TR::to_f32(b2[n]) : 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) s += red[w][n][r];
      y[mr * N + n] = TR::from_f32(s);
    }
  }
}

// grid = (k tiles of 256 hidden units, row chunks).  partial layout per chunk:
//   [n * H + k] n < NT : dW2      [NT * H + k] : dw1      [(NT + 1) * H + k] : db1      [(NT + 2) * H + n] : db2
template <typename TR, int NT, int ACT, bool WITH_DB2>
__device__ __forceinline__ void rank1_mlp_bwd_body(const typename TR::T* __restrict__ dy, const typename TR::T* __restrict__ x,
                                                   const typename TR::T* __restrict__ w1, const typename TR::T* __restrict__ b1,
                                                   const typename TR::T* __restrict__ W2, float* __restrict__ out, long long m0,
                                                   long long m1, int H, int N) {
  const int k = blockIdx.x * kThreads + threadIdx.x;
  const bool live = k < H;
  const int kc = live ? k : H - 1;
  const float w1k = TR::to_f32(w1[kc]);
  const float b1k = b1 ? TR::to_f32(b1[kc]) : 0.f;
  float w2k[NT], aW2[NT], adb2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    w2k[n] = n < N ? TR::to_f32(W2[static_cast<long long>(n) * H + kc]) : 0.f;
    aW2[n] = 0.f;
    adb2[n] = 0.f;
  }
  float aw1 = 0.f, ab1 = 0.f;
#pragma unroll 4
  for (long long m = m0; m < m1; ++m) {
    const float xv = TR::to_f32(x[m]);  // wave-uniform
    float g[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) g[n] = n < N ? TR::to_f32(dy[m * N + n]) : 0.f;  // wave-uniform
    const float pre = fmaf(xv, w1k, b1k);
    const float h = hidden_act<ACT>(pre);
    float dh = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      dh = fmaf(g[n], w2k[n], dh);
      aW2[n] = fmaf(g[n], h, aW2[n]);
      if constexpr (WITH_DB2) adb2[n] += g[n];
    }
    if constexpr (ACT == kActRelu) dh = pre > 0.f ? dh : 0.f;
    aw1 = fmaf(dh, xv, aw1);
    ab1 += dh;
  }
  if (live) {
#pragma unroll
    for (int n = 0; n < NT; ++n) out[static_cast<long long>(n) * H + k] = aW2[n];
    out[static_cast<long long>(NT) * H + k] = aw1;
    out[static_cast<long long>(NT + 1) * H + k] = ab1;
  }
  if constexpr (WITH_DB2) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n) out[static_cast<long long>(NT + 2) * H + n] = adb2[n];
    }
  }
}

template <typename TR, int NT, int ACT>
__global__ __launch_bounds__(kThreads) void rank1_mlp_bwd_kernel(const typename TR::T* __restrict__ dy,
                                                                 const typename TR::T* __restrict__ x,
                                                                 const typename TR::T* __restrict__ w1,
                                                                 const typename TR::T* __restrict__ b1,
                                                                 const typename TR::T* __restrict__ W2,
                                                                 float* __restrict__ partial, long long M, int H, int N,
                                                                 int rows_per_chunk) {
  const long long m0 = static_cast<long long>(blockIdx.y) * rows_per_chunk;
  const long long m1 = m0 + rows_per_chunk < M ? m0 + rows_per_chunk : M;
  float* out = partial + static_cast<long long>(blockIdx.y) * (static_cast<long long>(NT + 2) * H + NT);
  if (blockIdx.x == 0) rank1_mlp_bwd_body<TR, NT, ACT, true>(dy, x, w1, b1, W2, out, m0, m1, H, N);
  else rank1_mlp_bwd_body<TR, NT, ACT, false>(dy, x, w1, b1, W2, out, m0, m1, H, N);
}

template <typename F>
void dispatch_nt_act(long long N, int act, F&& f) {
  auto with_act = [&](auto nt) {
    if (act == kActRelu) f(nt, std::integral_constant<int, kActRelu>{});
    else f(nt, std::integral_constant<int, kActNone>{});
  };
  if (N <= 4) with_act(std::integral_constant<int, 4>{});
  else if (N <= 8) with_act(std::integral_constant<int, 8>{});
  else with_act(std::integral_constant<int, 16>{});
}

int rank1_nt(long long N) { return N <= 4 ? 4 : (N <= 8 ? 8 : 16); }

}  // namespace

void launch_rank1_mlp_fwd(const void* x, const void* w1, const void* b1, const void* W2, const void* b2, void* y, long long M,
                          long long H, long long N, int dtype, int act, hipStream_t stream) {
  if (M <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    dispatch_nt_act(N, act, [&](auto nt, auto a) {
      constexpr int NT = decltype(nt)::value, ACT = decltype(a)::value;
      const unsigned blocks = static_cast<unsigned>((M + 63) / 64);
      hipLaunchKernelGGL((rank1_mlp_fwd_kernel<TR, NT, ACT>), dim3(blocks), dim3(kThreads), 0, stream, static_cast<const T*>(x),
                         static_cast<const T*>(w1), static_cast<const T*>(b1), static_cast<const T*>(W2),
                         static_cast<const T*>(b2), static_cast<T*>(y), M, static_cast<int>(H), static_cast<int>(N));
    });
  });
}

int rank1_mlp_bwd_chunks(long long M, long long H) {
  // enough workgroups to fill the chip several times over: k tiles x chunks >= ~2048, chunks of >= 64 rows
  const long long ktiles = (H + kThreads - 1) / kThreads;
  long long chunks = (2048 + ktiles - 1) / ktiles;
  const long long max_chunks = (M + 63) / 64;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  if (chunks > 1024) chunks = 1024;
  return static_cast<int>(chunks);
}

long long rank1_mlp_bwd_len(long long H, long long N) {
  const int NT = rank1_nt(N);
  return static_cast<long long>(NT + 2) * H + NT;
}

// grads (fp32 or the parameter dtype, length rank1_mlp_bwd_len) = fixed-order sum of the chunk partials.
void launch_rank1_mlp_bwd(const void* dy, const void* x, const void* w1, const void* b1, const void* W2, float* partial,
                          void* grads, long long M, long long H, long long N, int dtype, int act, hipStream_t stream) {
  if (M <= 0) return;
  const int chunks = rank1_mlp_bwd_chunks(M, H);
  const int rpc = static_cast<int>((M + chunks - 1) / chunks);
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    dispatch_nt_act(N, act, [&](auto nt, auto a) {
      constexpr int NT = decltype(nt)::value, ACT = decltype(a)::value;
      dim3 grid(static_cast<unsigned>((H + kThreads - 1) / kThreads), static_cast<unsigned>(chunks));
      hipLaunchKernelGGL((rank1_mlp_bwd_kernel<TR, NT, ACT>), grid, dim3(kThreads), 0, stream, static_cast<const T*>(dy),
                         static_cast<const T*>(x), static_cast<const T*>(w1), static_cast<const T*>(b1),
                         static_cast<const T*>(W2), partial, M, static_cast<int>(H), static_cast<int>(N), rpc);
    });
  });
  launch_reduce_partials(partial, grads, rank1_mlp_bwd_len(H, N), chunks, dtype, stream);
}

}  // namespace kernels
}  // namespace dpa
