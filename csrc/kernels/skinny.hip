// Degenerate-shape linear layers and column reductions for gfx950.
//
// The reference model is Linear(1, H) -> Linear(H, 4) with no activation
// (reference min_DDP.py:41-49), so none of its six matrix products is GEMM-shaped:
//   forward   h = x w1^T + b1        K = 1   outer product        (outer_bias_act)
//             y = h W2^T + b2        N = 4   skinny output        (skinny_n)
//   backward  dh = dy W2             N = 4   broadcast-FMA        (skinny_n_dgrad)
//             dW2 = dy^T h, db2      N = 4   reduction over rows  (skinny_n_wgrad)
//             dw1 = dh^T x, db1      K = 1   weighted column sum  (col_reduce, kOuterWgrad)
// A 128x128 MFMA tile would be >96 % padding on every one of them, so they are written as
// what they are: HBM streams over the [M, H] activation with 16-byte accesses per lane,
// fp32 accumulation, and deterministic two-stage reductions (per-row-chunk partials in a
// workspace, then a second pass) instead of float atomics.
// The same column-reduction kernel provides bias gradients and the fused
// "activation backward + bias gradient" step of the large-GEMM path.
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == kActRelu) return v > 0.f ? v : 0.f;
  if (act == kActGelu) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  return v;
}

// ------------------------------------------------------------------ K = 1 forward
template <typename TR>
__global__ __launch_bounds__(kThreads) void outer_bias_act_kernel(const typename TR::T* __restrict__ x,
                                                                  const typename TR::T* __restrict__ w,
                                                                  const typename TR::T* __restrict__ bias,
                                                                  typename TR::T* __restrict__ y, long long M, long long N,
                                                                  int act, int rows_per_block) {
  constexpr int V = TR::kVec;
  // rows ride on grid.x (2^31 limit); grid.y (65535 limit) only counts column tiles
  const long long n0 = (static_cast<long long>(blockIdx.y) * kThreads + threadIdx.x) * V;
  if (n0 >= N) return;
  const long long m0 = static_cast<long long>(blockIdx.x) * rows_per_block;
  const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  const bool vec = (n0 + V <= N) && (N % V == 0) &&
                   ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0;
  if (vec) {
    float wv[V], bv[V];
    load_as_f32<TR, V>(w + n0, wv);
    if (bias) load_as_f32<TR, V>(bias + n0, bv);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) bv[j] = 0.f;
    }
    for (long long m = m0; m < m1; ++m) {
      const float xv = TR::to_f32(x[m]);
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = act_apply(fmaf(xv, wv[j], bv[j]), act);
      store_from_f32<TR, V>(y + m * N + n0, o);
    }
  } else {
    for (long long m = m0; m < m1; ++m) {
      const float xv = TR::to_f32(x[m]);
      for (int j = 0; j < V && n0 + j < N; ++j) {
        const float b = bias ? TR::to_f32(bias[n0 + j]) : 0.f;
        y[m * N + n0 + j] = TR::from_f32(act_apply(fmaf(xv, TR::to_f32(w[n0 + j]), b), act));
      }
    }
  }
}

// ------------------------------------------------------------------ N <= 16 forward
// One wave owns R rows; per step every lane takes V consecutive k of each row and of each
// weight row, so a weight vector fetched once serves R rows.
template <typename TR, int NT, int R>
__global__ __launch_bounds__(kThreads) void skinny_n_kernel(const typename TR::T* __restrict__ x,
                                                            const typename TR::T* __restrict__ w,
                                                            const typename TR::T* __restrict__ bias,
                                                            typename TR::T* __restrict__ y, long long M, int N, long long K) {
  constexpr int V = TR::kVec;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long m0 = (static_cast<long long>(blockIdx.x) * (kThreads / 64) + wid) * R;
  if (m0 >= M) return;
  float acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[r][n] = 0.f;

  const bool vec = (K % V == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
  if (vec) {
    for (long long k = static_cast<long long>(lane) * V; k < K; k += 64 * V) {
      float wv[NT][V];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n < N) load_as_f32<TR, V>(w + n * K + k, wv[n]);
        else {
#pragma unroll
          for (int j = 0; j < V; ++j) wv[n][j] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long long m = m0 + r < M ? m0 + r : M - 1;  // clamp: tail rows recompute the last row
        float xv[V];
        load_as_f32<TR, V>(x + m * K + k, xv);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int j = 0; j < V; ++j) acc[r][n] = fmaf(xv[j], wv[n][j], acc[r][n]);
      }
    }
  } else {
    for (long long k = lane; k < K; k += 64) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long long m = m0 + r < M ? m0 + r : M - 1;
        const float xv = TR::to_f32(x[m * K + k]);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (n < N) acc[r][n] = fmaf(xv, TR::to_f32(w[n * K + k]), acc[r][n]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[r][n] = wave_sum(acc[r][n]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (m0 + r >= M) break;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        if (n < N) y[(m0 + r) * N + n] = TR::from_f32(acc[r][n] + (bias ? TR::to_f32(bias[n]) : 0.f));
    }
  }
}

// ------------------------------------------------------------------ N <= 16 input gradient
template <typename TR, int NT>
__global__ __launch_bounds__(kThreads) void skinny_n_dgrad_kernel(const typename TR::T* __restrict__ dy,
                                                                  const typename TR::T* __restrict__ w,
                                                                  typename TR::T* __restrict__ dx, long long M, int N,
                                                                  long long K, int rows_per_block) {
  constexpr int V = TR::kVec;
  const long long k0 = (static_cast<long long>(blockIdx.y) * kThreads + threadIdx.x) * V;
  if (k0 >= K) return;
  const long long m0 = static_cast<long long>(blockIdx.x) * rows_per_block;
  const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  const bool vec = (K % V == 0) && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
  float wv[NT][V];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (n < N && vec) load_as_f32<TR, V>(w + n * K + k0, wv[n]);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) wv[n][j] = (n < N && k0 + j < K) ? TR::to_f32(w[n * K + k0 + j]) : 0.f;
    }
  }
  for (long long m = m0; m < m1; ++m) {
    float o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (n < N) {
        const float g = TR::to_f32(dy[m * N + n]);  // same address in every lane: one broadcast fetch
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = fmaf(g, wv[n][j], o[j]);
      }
    }
    if (vec) store_from_f32<TR, V>(dx + m * K + k0, o);
    else
      for (int j = 0; j < V && k0 + j < K; ++j) dx[m * K + k0 + j] = TR::from_f32(o[j]);
  }
}

// ------------------------------------------------------------------ N <= 16 weight gradient, stage 1
// grid = (column tiles, row chunks); partial[chunk][n][k] = sum over the chunk's rows of dy[m][n] x[m][k]
template <typename TR, int NT>
__global__ __launch_bounds__(kThreads) void skinny_n_wgrad_partial_kernel(const typename TR::T* __restrict__ dy,
                                                                          const typename TR::T* __restrict__ x,
                                                                          float* __restrict__ partial, long long M, int N,
                                                                          long long K, int rows_per_chunk) {
  constexpr int V = TR::kVec;
  const long long k0 = (static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x) * V;
  if (k0 >= K) return;
  const long long m0 = static_cast<long long>(blockIdx.y) * rows_per_chunk;
  const long long m1 = m0 + rows_per_chunk < M ? m0 + rows_per_chunk : M;
  const bool vec = (K % V == 0) && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  float acc[NT][V];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[n][j] = 0.f;
  for (long long m = m0; m < m1; ++m) {
    float xv[V];
    if (vec) load_as_f32<TR, V>(x + m * K + k0, xv);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) xv[j] = (k0 + j < K) ? TR::to_f32(x[m * K + k0 + j]) : 0.f;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (n < N) {
        const float g = TR::to_f32(dy[m * N + n]);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[n][j] = fmaf(g, xv[j], acc[n][j]);
      }
    }
  }
  float* out = partial + static_cast<long long>(blockIdx.y) * N * K;
#pragma unroll
  for (int n = 0; n < NT; ++n)
    if (n < N)
      for (int j = 0; j < V && k0 + j < K; ++j) out[n * K + k0 + j] = acc[n][j];
}

// out[i] = sum_c partial[c][i], i in [0, len).  A workgroup owns 16 consecutive outputs; its 256
// threads form 16 row groups x 16 columns (64-byte row segments), each thread keeps four
// independent loads in flight while it walks chunks/16 rows (the kernel is latency-bound), and LDS
// combines the 16 row groups.
template <typename TR>
__global__ __launch_bounds__(kThreads) void reduce_partials_kernel(const float* __restrict__ partial,
                                                                   typename TR::T* __restrict__ out, long long len,
                                                                   int chunks) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const long long i = static_cast<long long>(blockIdx.x) * 16 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < len) {
    int r = rg;
    for (; r + 48 < chunks; r += 64) {
      s0 += partial[static_cast<long long>(r) * len + i];
      s1 += partial[static_cast<long long>(r + 16) * len + i];
      s2 += partial[static_cast<long long>(r + 32) * len + i];
      s3 += partial[static_cast<long long>(r + 48) * len + i];
    }
    for (; r < chunks; r += 16) s0 += partial[static_cast<long long>(r) * len + i];
  }
  red[rg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && i < len) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    out[i] = TR::from_f32(t);
  }
}

// The same sum for thousands of partial rows (per-32-row column sums of an attention gradient: 12288 x 2304
// values): there the kernel above, with its 64-byte row pieces and len / 16 small workgroups, reads at a tenth
// of the memory rate.  Here a workgroup of 1024 threads owns 32 consecutive outputs: 64 row groups x 16 lanes
// of 8 bytes, i.e. one full 128-byte line per row, four independent loads in flight per thread; the row groups
// are combined through LDS in a fixed order.  Needs an even `len`.
template <typename TR>
__global__ __launch_bounds__(1024) void reduce_partials_wide_kernel(const float* __restrict__ partial, typename TR::T* __restrict__ out,
                                                                    long long len, int chunks) {
  __shared__ float red[64][33];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const long long i = static_cast<long long>(blockIdx.x) * 32 + 2 * c;
  float2 s0 = {0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (i < len) {
    const float* p = partial + i;
    auto at = [&](int r) { return *reinterpret_cast<const float2*>(p + static_cast<long long>(r) * len); };
    int r = rg;
    for (; r + 192 < chunks; r += 256) {
      const float2 a = at(r), b = at(r + 64), e = at(r + 128), f = at(r + 192);
      s0.x += a.x; s0.y += a.y;
      s1.x += b.x; s1.y += b.y;
      s2.x += e.x; s2.y += e.y;
      s3.x += f.x; s3.y += f.y;
    }
    for (; r < chunks; r += 64) {
      const float2 a = at(r);
      s0.x += a.x; s0.y += a.y;
    }
  }
  red[rg][2 * c] = (s0.x + s1.x) + (s2.x + s3.x);
  red[rg][2 * c + 1] = (s0.y + s1.y) + (s2.y + s3.y);
  __syncthreads();
  if (threadIdx.x < 32) {
    const long long o = static_cast<long long>(blockIdx.x) * 32 + threadIdx.x;
    if (o < len) {
      float t = 0.f;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) t += red[k][threadIdx.x];
      out[o] = TR::from_f32(t);
    }
  }
}

template <typename TR>
void reduce_partials(const float* partial, typename TR::T* out, long long len, int chunks, hipStream_t stream) {
  if (chunks >= 512 && (len & 1) == 0 && (reinterpret_cast<uintptr_t>(partial) & 7) == 0)
    hipLaunchKernelGGL((reduce_partials_wide_kernel<TR>), dim3(static_cast<unsigned>((len + 31) / 32)), dim3(1024), 0, stream, partial, out, len, chunks);
  else
    hipLaunchKernelGGL((reduce_partials_kernel<TR>), dim3(static_cast<unsigned>((len + 15) / 16)), dim3(kThreads), 0, stream, partial, out, len, chunks);
}

// ------------------------------------------------------------------ column reductions over [M, N]
enum ColMode : int { kColSum = 0, kOuterWgrad = 1, kActBwd = 2 };

// grid = (column tiles, row chunks).  Per mode:
//   kColSum     : partial_b[chunk][n] = sum_m dy[m][n]
//   kOuterWgrad : also partial_w[chunk][n] = sum_m dy[m][n] * xrow[m]
//   kActBwd     : g[m][n] = dy[m][n] * act'(saved[m][n]) is written, partial_b sums g
// A workgroup is 8 row lanes x 32 column lanes; a column lane owns one 16-byte vector, so a
// column tile is 512 bytes of a row (fully used cache lines) and even a 768-column matrix keeps
// every lane busy.  Each thread walks rows m0 + r, m0 + r + 8, ...; LDS folds the 8 row lanes.
constexpr int kColLanes = 32, kRowLanes = 8;

template <typename TR, int MODE>
__global__ __launch_bounds__(kThreads) void col_reduce_partial_kernel(const typename TR::T* __restrict__ dy,
                                                                      const typename TR::T* __restrict__ aux,
                                                                      typename TR::T* __restrict__ g_out,
                                                                      float* __restrict__ partial_b,
                                                                      float* __restrict__ partial_w, long long M,
                                                                      long long N, int rows_per_chunk, int act) {
  constexpr int V = TR::kVec;
  constexpr int TILE = kColLanes * V;
  __shared__ float red[kRowLanes][TILE + 1];
  const int cl = threadIdx.x & (kColLanes - 1), rl = threadIdx.x / kColLanes;
  const long long n0 = (static_cast<long long>(blockIdx.x) * kColLanes + cl) * V;
  const long long m0 = static_cast<long long>(blockIdx.y) * rows_per_chunk;
  const long long m1 = m0 + rows_per_chunk < M ? m0 + rows_per_chunk : M;
  uintptr_t bits = reinterpret_cast<uintptr_t>(dy);
  if (MODE == kActBwd) bits |= reinterpret_cast<uintptr_t>(aux) | reinterpret_cast<uintptr_t>(g_out);
  const bool vec = (N % V == 0) && (bits & 15) == 0;
  float sb[V], sw[V];
#pragma unroll
  for (int j = 0; j < V; ++j) sb[j] = sw[j] = 0.f;
  if (n0 < N) {
#pragma unroll 4
    for (long long m = m0 + rl; m < m1; m += kRowLanes) {
      float d[V];
      if (vec) load_as_f32<TR, V>(dy + m * N + n0, d);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = (n0 + j < N) ? TR::to_f32(dy[m * N + n0 + j]) : 0.f;
      }
      if (MODE == kActBwd) {
        float sv[V];
        if (vec) load_as_f32<TR, V>(aux + m * N + n0, sv);
        else {
#pragma unroll
          for (int j = 0; j < V; ++j) sv[j] = (n0 + j < N) ? TR::to_f32(aux[m * N + n0 + j]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
          if (act == kActRelu) d[j] = sv[j] > 0.f ? d[j] : 0.f;
          else if (act == kActGelu) d[j] *= gelu_grad(sv[j]);
        }
        if (vec) store_from_f32<TR, V>(g_out + m * N + n0, d);
        else
          for (int j = 0; j < V && n0 + j < N; ++j) g_out[m * N + n0 + j] = TR::from_f32(d[j]);
      }
      if (MODE == kOuterWgrad) {
        const float xv = TR::to_f32(aux[m]);
#pragma unroll
        for (int j = 0; j < V; ++j) sw[j] = fmaf(d[j], xv, sw[j]);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) sb[j] += d[j];
    }
  }
  // fold the 8 row lanes: thread t then owns column t of the tile
  const long long nt = static_cast<long long>(blockIdx.x) * TILE + threadIdx.x;
#pragma unroll
  for (int j = 0; j < V; ++j) red[rl][cl * V + j] = sb[j];
  __syncthreads();
  if (threadIdx.x < TILE && nt < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < kRowLanes; ++r) t += red[r][threadIdx.x];
    partial_b[static_cast<long long>(blockIdx.y) * N + nt] = t;
  }
  if (MODE == kOuterWgrad) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < V; ++j) red[rl][cl * V + j] = sw[j];
    __syncthreads();
    if (threadIdx.x < TILE && nt < N) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < kRowLanes; ++r) t += red[r][threadIdx.x];
      partial_w[static_cast<long long>(blockIdx.y) * N + nt] = t;
    }
  }
}

// Column sum of a narrow matrix (N <= 16): a lane owns whole rows, so all 256 lanes work
// even when a row is only 16 bytes; partial_b[chunk][n] as above.
template <typename TR, int NT>
__global__ __launch_bounds__(kThreads) void colsum_small_partial_kernel(const typename TR::T* __restrict__ dy,
                                                                        float* __restrict__ partial_b, long long M, int N,
                                                                        int rows_per_chunk) {
  __shared__ float scratch[kThreads / 64];
  const long long m0 = static_cast<long long>(blockIdx.x) * rows_per_chunk;
  const long long m1 = m0 + rows_per_chunk < M ? m0 + rows_per_chunk : M;
  float acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = 0.f;
  for (long long m = m0 + threadIdx.x; m < m1; m += kThreads) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (n < N) acc[n] += TR::to_f32(dy[m * N + n]);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float s = block_sum<kThreads / 64>(acc[n], scratch);
    if (threadIdx.x == 0 && n < N) partial_b[static_cast<long long>(blockIdx.x) * N + n] = s;
  }
}

int pick_rows_per_chunk(long long M, int max_chunks) {
  long long r = (M + max_chunks - 1) / max_chunks;
  if (r < 32) r = 32;
  return static_cast<int>(r);
}

constexpr int kMaxChunks = 512;

template <typename TR, int MODE>
void col_reduce(const void* dy, const void* aux, void* g, void* db, void* dw, float* workspace, long long M, long long N,
                int act, hipStream_t stream) {
  using T = typename TR::T;
  const int rpc = pick_rows_per_chunk(M, kMaxChunks);
  const int chunks = static_cast<int>((M + rpc - 1) / rpc);
  const int col_tiles = static_cast<int>((N + kColLanes * TR::kVec - 1) / (kColLanes * TR::kVec));
  float* pb = workspace;
  float* pw = workspace + static_cast<long long>(chunks) * N;
  if (MODE == kColSum && N <= 16) {
    hipLaunchKernelGGL((colsum_small_partial_kernel<TR, 16>), dim3(chunks), dim3(kThreads), 0, stream,
                       static_cast<const T*>(dy), pb, M, static_cast<int>(N), rpc);
  } else
  hipLaunchKernelGGL((col_reduce_partial_kernel<TR, MODE>), dim3(col_tiles, chunks), dim3(kThreads), 0, stream,
                     static_cast<const T*>(dy), static_cast<const T*>(aux), static_cast<T*>(g), pb, pw, M, N, rpc, act);
  const int rb = static_cast<int>((N + 15) / 16);
  if (db)
    hipLaunchKernelGGL((reduce_partials_kernel<TR>), dim3(rb), dim3(kThreads), 0, stream, pb, static_cast<T*>(db), N, chunks);
  if (MODE == kOuterWgrad && dw)
    hipLaunchKernelGGL((reduce_partials_kernel<TR>), dim3(rb), dim3(kThreads), 0, stream, pw, static_cast<T*>(dw), N, chunks);
}

}  // namespace

void launch_outer_bias_act(const void* x, const void* w, const void* bias, void* y, long long M, long long N, int dtype,
                           int act, hipStream_t stream) {
  if (M <= 0 || N <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    const int rows = 8;
    dim3 grid(static_cast<unsigned>((M + rows - 1) / rows),
              static_cast<unsigned>((N + kThreads * TR::kVec - 1) / (kThreads * TR::kVec)));
    hipLaunchKernelGGL((outer_bias_act_kernel<TR>), grid, dim3(kThreads), 0, stream, static_cast<const T*>(x),
                       static_cast<const T*>(w), static_cast<const T*>(bias), static_cast<T*>(y), M, N, act, rows);
  });
}

void launch_skinny_n(const void* x, const void* w, const void* bias, void* y, long long M, long long N, long long K,
                     int dtype, hipStream_t stream) {
  if (M <= 0 || N <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    auto go = [&](auto nt, auto r) {
      constexpr int NT = decltype(nt)::value, R = decltype(r)::value;
      const long long rows_per_block = static_cast<long long>(kThreads / 64) * R;
      const unsigned blocks = static_cast<unsigned>((M + rows_per_block - 1) / rows_per_block);
      hipLaunchKernelGGL((skinny_n_kernel<TR, NT, R>), dim3(blocks), dim3(kThreads), 0, stream, static_cast<const T*>(x),
                         static_cast<const T*>(w), static_cast<const T*>(bias), static_cast<T*>(y), M, static_cast<int>(N), K);
    };
    if (N <= 4) go(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
    else if (N <= 8) go(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
    else go(std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
  });
}

void launch_skinny_n_dgrad(const void* dy, const void* w, void* dx, long long M, long long N, long long K, int dtype,
                           hipStream_t stream) {
  if (M <= 0 || K <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    const int rows = 8;
    dim3 grid(static_cast<unsigned>((M + rows - 1) / rows),
              static_cast<unsigned>((K + kThreads * TR::kVec - 1) / (kThreads * TR::kVec)));
    auto go = [&](auto nt) {
      constexpr int NT = decltype(nt)::value;
      hipLaunchKernelGGL((skinny_n_dgrad_kernel<TR, NT>), grid, dim3(kThreads), 0, stream, static_cast<const T*>(dy),
                         static_cast<const T*>(w), static_cast<T*>(dx), M, static_cast<int>(N), K, rows);
    };
    if (N <= 4) go(std::integral_constant<int, 4>{});
    else if (N <= 8) go(std::integral_constant<int, 8>{});
    else go(std::integral_constant<int, 16>{});
  });
}

long long skinny_wgrad_workspace(long long M, long long N, long long K) {
  const int rpc = pick_rows_per_chunk(M, kMaxChunks);
  const long long chunks = (M + rpc - 1) / rpc;
  return chunks * N * K + 2 * chunks * N;
}

void launch_skinny_n_wgrad(const void* dy, const void* x, void* dW, void* db, float* workspace, long long M, long long N,
                           long long K, int dtype, hipStream_t stream) {
  if (M <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    const int rpc = pick_rows_per_chunk(M, kMaxChunks);
    const int chunks = static_cast<int>((M + rpc - 1) / rpc);
    dim3 grid(static_cast<unsigned>((K + kThreads * TR::kVec - 1) / (kThreads * TR::kVec)), static_cast<unsigned>(chunks));
    auto go = [&](auto nt) {
      constexpr int NT = decltype(nt)::value;
      hipLaunchKernelGGL((skinny_n_wgrad_partial_kernel<TR, NT>), grid, dim3(kThreads), 0, stream, static_cast<const T*>(dy),
                         static_cast<const T*>(x), workspace, M, static_cast<int>(N), K, rpc);
    };
    if (N <= 4) go(std::integral_constant<int, 4>{});
    else if (N <= 8) go(std::integral_constant<int, 8>{});
    else go(std::integral_constant<int, 16>{});
    const long long len = N * K;
    hipLaunchKernelGGL((reduce_partials_kernel<TR>), dim3(static_cast<unsigned>((len + 15) / 16)),
                       dim3(kThreads), 0, stream, workspace, static_cast<T*>(dW), len, chunks);
    if (db) col_reduce<TR, kColSum>(dy, nullptr, nullptr, db, nullptr, workspace + static_cast<long long>(chunks) * len, M, N, 0, stream);
  });
}

long long colreduce_workspace(long long M, long long N) {
  const int rpc = pick_rows_per_chunk(M, kMaxChunks);
  const long long chunks = (M + rpc - 1) / rpc;
  return 2 * chunks * N;
}

void launch_outer_wgrad(const void* dy, const void* x, void* dw, void* db, float* workspace, long long M, long long N,
                        int dtype, hipStream_t stream) {
  if (M <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    col_reduce<TR, kOuterWgrad>(dy, x, nullptr, db, dw, workspace, M, N, 0, stream);
  });
}

void launch_reduce_partials(const float* partial, void* out, long long len, int chunks, int dtype, hipStream_t stream) {
  if (len <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    reduce_partials<TR>(partial, static_cast<typename TR::T*>(out), len, chunks, stream);
  });
}

void launch_colsum(const void* dy, void* db, float* workspace, long long M, long long N, int dtype, hipStream_t stream) {
  if (M <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    col_reduce<TR, kColSum>(dy, nullptr, nullptr, db, nullptr, workspace, M, N, 0, stream);
  });
}

void launch_act_bwd(const void* dy, const void* saved, void* g, void* db, float* workspace, long long M, long long N,
                    int dtype, int act, hipStream_t stream) {
  if (M <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    col_reduce<TR, kActBwd>(dy, saved, g, db, nullptr, workspace, M, N, act, stream);
  });
}

}  // namespace kernels
}  // namespace dpa
