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
constexpr int kFwdWaves = 8;  // forward: the workgroup's waves split the hidden units (and hide each other's scalar-load latency)
constexpr int kFwdThreads = kFwdWaves * 64;

// s_waitcnt lgkmcnt(0) as an instruction the compiler's own wait bookkeeping sees.  Scalar loads return out
// of order, so the only wait there is drains all of them: placed once in front of a prefetching loop, it
// lets the compiler put the loop's single wait where the prefetched values are first needed.
__device__ __forceinline__ void wait_scalar_loads() { __builtin_amdgcn_s_waitcnt(0xC07F); }

template <int ACT>
__device__ __forceinline__ float hidden_act(float h) {
  if constexpr (ACT == kActRelu) return h > 0.f ? h : 0.f;
  return h;
}

// Weights of the forward pass as one fp32 stream of records {w1[k], b1[k], W2[0][k], ..., W2[NT-1][k]}
// (rows of W2 beyond N and records beyond H are zero).  The forward kernel's operands are all
// wave-uniform; fetched from their own arrays they cost one scalar load plus 64-bit address arithmetic
// per operand and hidden unit (2.3 scalar instructions per vector one, measured), and pairs of W2 values
// that a packed FMA wants side by side sit H elements apart.  From the packed stream a wave fetches two
// hidden units' operands with 16-byte scalar loads off one running pointer.
template <typename TR, int NT>
__global__ __launch_bounds__(kThreads) void rank1_pack_kernel(const typename TR::T* __restrict__ w1,
                                                              const typename TR::T* __restrict__ b1,
                                                              const typename TR::T* __restrict__ W2, float* __restrict__ pk,
                                                              int H, int N, int records) {
  constexpr int R = NT + 2;
  const int k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= records) return;
  float* r = pk + static_cast<long long>(k) * R;
  const bool live = k < H;
  r[0] = live ? TR::to_f32(w1[k]) : 0.f;
  r[1] = (live && b1) ? TR::to_f32(b1[k]) : 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n) r[2 + n] = (live && n < N) ? TR::to_f32(W2[static_cast<long long>(n) * H + k]) : 0.f;
}

// grid.x = row blocks of 64 rows; wave w of the workgroup covers records [w * kq, (w + 1) * kq), kq even.
template <typename TR, int NT, int ACT>
__global__ __launch_bounds__(kFwdThreads) void rank1_mlp_fwd_kernel(const typename TR::T* __restrict__ x,
                                                                 const float* __restrict__ pk,
                                                                 const typename TR::T* __restrict__ b2,
                                                                 typename TR::T* __restrict__ y, long long M, int kq, int N) {
  constexpr int R = NT + 2;
  __shared__ float red[kFwdWaves][NT][64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long m = static_cast<long long>(blockIdx.x) * 64 + lane;
  const float xv = m < M ? TR::to_f32(x[m]) : 0.f;
  float acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = 0.f;
  const f32x4* p = reinterpret_cast<const f32x4*>(pk + static_cast<long long>(wid) * kq * R);  // 16-byte aligned: kq * R % 4 == 0
  constexpr int Q = 2 * R / 4;  // 16-byte pieces per pair of records
  // Scalar loads return out of order, so the one wait there is drains all of them and a software prefetch
  // buys nothing; the latency is hidden by the workgroup's eight waves instead.
#pragma unroll 2
  for (int k = 0; k < kq; k += 2) {
    float rec[2 * R];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const f32x4 v = p[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) rec[4 * q + j] = v[j];
    }
    p += Q;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const float h = hidden_act<ACT>(fmaf(xv, rec[h2 * R], rec[h2 * R + 1]));
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = fmaf(h, rec[h2 * R + 2 + n], acc[n]);
    }
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) red[wid][n][lane] = acc[n];
  __syncthreads();
  // 64 rows x NT outputs, folded over the four k-quarters in a fixed order
  for (int i = threadIdx.x; i < 64 * NT; i += kFwdThreads) {
    const int r = i / NT, n = i % NT;
    const long long mr = static_cast<long long>(blockIdx.x) * 64 + r;
    if (n < N && mr < M) {
      float s = b2 ? TR::to_f32(b2[n]) : 0.f;
#pragma unroll
      for (int w = 0; w < kFwdWaves; ++w) s += red[w][n][r];
      y[mr * N + n] = TR::from_f32(s);
    }
  }
}

// grid = (k tiles of 512 hidden units, row chunks).  partial layout per chunk:
//   [n * H + k] n < NT : dW2      [NT * H + k] : dw1      [(NT + 1) * H + k] : db1      [(NT + 2) * H + n] : db2
typedef __attribute__((ext_vector_type(2))) float f32x2;

// A lane owns TWO neighbouring hidden units, held as 2-vectors, so that every operation of a row is one
// packed FMA (v_pk_fma_f32 with the wave-uniform x / dy value broadcast from a scalar register): the
// kernel is bound by vector issue, and packed math halves the instructions per (row, hidden unit).
template <typename TR, int NT, int ACT, bool WITH_DB2>
__device__ __forceinline__ void rank1_mlp_bwd_body(const typename TR::T* __restrict__ dy, const typename TR::T* __restrict__ x,
                                                   const typename TR::T* __restrict__ w1, const typename TR::T* __restrict__ b1,
                                                   const typename TR::T* __restrict__ W2, float* __restrict__ out, long long m0,
                                                   long long m1, int H, int N) {
  const int k = (blockIdx.x * kThreads + threadIdx.x) * 2;
  const bool live0 = k < H, live1 = k + 1 < H;
  const int kc0 = live0 ? k : H - 1, kc1 = live1 ? k + 1 : H - 1;
  const f32x2 w1k = {TR::to_f32(w1[kc0]), TR::to_f32(w1[kc1])};
  const f32x2 b1k = b1 ? f32x2{TR::to_f32(b1[kc0]), TR::to_f32(b1[kc1])} : f32x2{0.f, 0.f};
  f32x2 w2k[NT], aW2[NT];
  float adb2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const long long r = static_cast<long long>(n) * H;
    w2k[n] = n < N ? f32x2{TR::to_f32(W2[r + kc0]), TR::to_f32(W2[r + kc1])} : f32x2{0.f, 0.f};
    aW2[n] = f32x2{0.f, 0.f};
    adb2[n] = 0.f;
  }
  f32x2 aw1 = {0.f, 0.f}, ab1 = {0.f, 0.f};
  auto row = [&](float xs, const float (&g)[NT]) __attribute__((always_inline)) {
    const f32x2 xv = {xs, xs};
    const f32x2 pre = xv * w1k + b1k;
    f32x2 h = pre;
    if constexpr (ACT == kActRelu) h = f32x2{pre[0] > 0.f ? pre[0] : 0.f, pre[1] > 0.f ? pre[1] : 0.f};
    f32x2 dh = {0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const f32x2 gv = {g[n], g[n]};
      dh = gv * w2k[n] + dh;
      aW2[n] = gv * h + aW2[n];
      if constexpr (WITH_DB2) adb2[n] += g[n];
    }
    if constexpr (ACT == kActRelu) dh = f32x2{pre[0] > 0.f ? dh[0] : 0.f, pre[1] > 0.f ? dh[1] : 0.f};
    aw1 = dh * xv + aw1;
    ab1 += dh;
  };
  long long m = m0;
  if constexpr (std::is_same<TR, F32>::value && NT == 4) {
    // Four rows per step: x[m..m+3] and the rows' 4 x 4 output gradients arrive as five 16-byte
    // wave-uniform loads instead of twenty scalar ones.  (Requesting the next four rows ahead of the
    // math does not help: scalar loads return out of order, so the compiler's only wait drains them all.)
    if (N == 4 && ((reinterpret_cast<uintptr_t>(x + m0) | reinterpret_cast<uintptr_t>(dy + m0 * 4)) & 15) == 0) {
#pragma unroll 2
      for (; m + 4 <= m1; m += 4) {
        const f32x4 xs = *reinterpret_cast<const f32x4*>(x + m);
        f32x4 gs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gs[r] = *reinterpret_cast<const f32x4*>(dy + (m + r) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g[NT] = {gs[r][0], gs[r][1], gs[r][2], gs[r][3]};
          row(xs[r], g);
        }
      }
    }
  }
#pragma unroll 2
  for (; m < m1; ++m) {
    float g[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) g[n] = n < N ? TR::to_f32(dy[m * N + n]) : 0.f;  // wave-uniform
    row(TR::to_f32(x[m]), g);                                                      // x[m]: wave-uniform
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c == 0 ? live0 : live1) {
#pragma unroll
      for (int n = 0; n < NT; ++n) out[static_cast<long long>(n) * H + k + c] = aW2[n][c];
      out[static_cast<long long>(NT) * H + k + c] = aw1[c];
      out[static_cast<long long>(NT + 1) * H + k + c] = ab1[c];
    }
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

long long rank1_mlp_fwd_workspace(long long H, long long N) {
  const long long kq = ((H + kFwdWaves - 1) / kFwdWaves + 1) / 2 * 2;
  return kFwdWaves * kq * (rank1_nt(N) + 2);
}

// pk: rank1_mlp_fwd_workspace(H, N) floats of scratch.
void launch_rank1_mlp_fwd(const void* x, const void* w1, const void* b1, const void* W2, const void* b2, void* y, float* pk,
                          long long M, long long H, long long N, int dtype, int act, hipStream_t stream) {
  if (M <= 0) return;
  const int kq = static_cast<int>(((H + kFwdWaves - 1) / kFwdWaves + 1) / 2 * 2);
  const int records = kFwdWaves * kq;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    dispatch_nt_act(N, act, [&](auto nt, auto a) {
      constexpr int NT = decltype(nt)::value, ACT = decltype(a)::value;
      hipLaunchKernelGGL((rank1_pack_kernel<TR, NT>), dim3((records + kThreads - 1) / kThreads), dim3(kThreads), 0, stream,
                         static_cast<const T*>(w1), static_cast<const T*>(b1), static_cast<const T*>(W2), pk,
                         static_cast<int>(H), static_cast<int>(N), records);
      const unsigned blocks = static_cast<unsigned>((M + 63) / 64);
      hipLaunchKernelGGL((rank1_mlp_fwd_kernel<TR, NT, ACT>), dim3(blocks), dim3(kFwdThreads), 0, stream, static_cast<const T*>(x),
                         static_cast<const float*>(pk), static_cast<const T*>(b2), static_cast<T*>(y), M, kq, static_cast<int>(N));
    });
  });
}

int rank1_mlp_bwd_chunks(long long M, long long H) {
  // enough workgroups to fill the chip several times over: k tiles x chunks >= ~2048, chunks of >= 64 rows
  const long long ktiles = (H + 2 * kThreads - 1) / (2 * kThreads);
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
      dim3 grid(static_cast<unsigned>((H + 2 * kThreads - 1) / (2 * kThreads)), static_cast<unsigned>(chunks));
      hipLaunchKernelGGL((rank1_mlp_bwd_kernel<TR, NT, ACT>), grid, dim3(kThreads), 0, stream, static_cast<const T*>(dy),
                         static_cast<const T*>(x), static_cast<const T*>(w1), static_cast<const T*>(b1),
                         static_cast<const T*>(W2), partial, M, static_cast<int>(H), static_cast<int>(N), rpc);
    });
  });
  launch_reduce_partials(partial, grads, rank1_mlp_bwd_len(H, N), chunks, dtype, stream);
}

}  // namespace kernels
}  // namespace dpa
