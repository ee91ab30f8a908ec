// Training-mode batch normalisation over a channels-last activation, i.e. over the rows of a row-major
// [R, C] matrix (R = batch * height * width positions, C channels), with the ReLU and the residual add of a
// ResNet block folded in.
//
// Not an op of the reference (its model is two linear layers, reference min_DDP.py:41-49); it exists for the
// ResNet-50 data-parallel configuration of this framework's benchmark set, where the stock channels-last
// batch-norm kernels take 50 of the 80 ms of a training step at batch 256 (per step they move about 45 GB,
// i.e. ~9 ms at HBM speed: the per-channel reductions are what is slow there).
//
// Forward:   statistics (one read of x)            bn_stats_partial_kernel -> bn_stats_finalize_kernel
//            y = relu(x * a[c] + b[c] (+ residual))  bn_apply_kernel, a = gamma * rstd, b = beta - mean * a
// Backward:  g = dy masked by y > 0 (when the ReLU is fused);  dbeta = sum g, dgamma = sum g * xhat
//                                                   bn_bwd_partial_kernel -> bn_bwd_finalize_kernel
//            dx = (g - dbeta / R - xhat * dgamma / R) * a  (and d residual = g)   bn_bwd_apply_kernel
//
// Reductions over rows are two-stage and fixed-order (a partial per row chunk, then a fixed tree over the chunks
// per channel), so results are bitwise reproducible.  The variance is accumulated as
// (count, mean, M2) triples merged with Chan's formula; inside a thread the sums are taken relative to the
// thread's first element, so a large common offset of a channel does not cancel away the variance.
// Lanes read 16 bytes of consecutive channels; a workgroup covers up to 256 such vectors of a row and
// 256 / (vectors per row) rows per pass.
#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

constexpr int kBnThreads = 256;

// Per-channel vectors (affine parameters, running statistics and their gradients) come in whatever dtype the
// module holds them in -- fp32, or the activation dtype after model.to(bfloat16): read / written per element.
__device__ __forceinline__ float chan_load(const void* p, long long c, int dt) {
  if (dt == kF32) return static_cast<const float*>(p)[c];
  const u16 raw = static_cast<const u16*>(p)[c];
  return dt == kBF16 ? BF16::to_f32(raw) : F16::to_f32(raw);
}
__device__ __forceinline__ void chan_store(void* p, long long c, int dt, float v) {
  if (dt == kF32) static_cast<float*>(p)[c] = v;
  else static_cast<u16*>(p)[c] = dt == kBF16 ? BF16::from_f32(v) : F16::from_f32(v);
}

// Geometry shared by the two reduction kernels: blockIdx.x = row chunk, blockIdx.y = block of 256 vectors.
struct BnTile {
  int cw;       // vectors of a row this workgroup covers (<= 256)
  int rpp;      // rows per pass
  int rl, cl;   // this thread's row lane and vector lane
  long long col0;  // first channel of this thread's vector
  bool active;
};
template <int V>
__device__ __forceinline__ BnTile bn_tile(long long C) {
  BnTile t;
  const long long cv = C / V;
  const long long first = static_cast<long long>(blockIdx.y) * kBnThreads;
  const long long left = cv - first;
  t.cw = static_cast<int>(left < kBnThreads ? left : kBnThreads);
  t.rpp = kBnThreads / t.cw;
  t.rl = threadIdx.x / t.cw;
  t.cl = threadIdx.x - t.rl * t.cw;
  t.active = t.rl < t.rpp;
  t.col0 = (first + t.cl) * V;
  return t;
}

// partial[chunk][0][c] = mean, partial[chunk][1][c] = M2 of the chunk's rows (count = rows in the chunk).
template <typename TR>
__global__ __launch_bounds__(kBnThreads) void bn_stats_partial_kernel(const typename TR::T* __restrict__ x, long long R, long long C,
                                                                      long long rows_per_chunk, float* __restrict__ partial) {
  constexpr int V = TR::kVec;
  __shared__ float s_mean[kBnThreads * V], s_m2[kBnThreads * V];
  __shared__ int s_n[kBnThreads];
  const BnTile t = bn_tile<V>(C);
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_chunk;
  const long long r1 = r0 + rows_per_chunk < R ? r0 + rows_per_chunk : R;
  float s[V], q[V], k[V];
  int n = 0;
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = q[j] = k[j] = 0.f;
  if (t.active) {
    long long r = r0 + t.rl;
    if (r < r1) load_as_f32<TR, V>(x + r * C + t.col0, k);  // the shift: this thread's first element of every channel
    // two rows in flight per trip
    for (; r + t.rpp < r1; r += 2 * t.rpp) {
      float a[V], b[V];
      load_as_f32<TR, V>(x + r * C + t.col0, a);
      load_as_f32<TR, V>(x + (r + t.rpp) * C + t.col0, b);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float da = a[j] - k[j], db = b[j] - k[j];
        s[j] += da + db;
        q[j] = fmaf(da, da, fmaf(db, db, q[j]));
      }
      n += 2;
    }
    for (; r < r1; r += t.rpp) {
      float a[V];
      load_as_f32<TR, V>(x + r * C + t.col0, a);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float da = a[j] - k[j];
        s[j] += da;
        q[j] = fmaf(da, da, q[j]);
      }
      n += 1;
    }
  }
  const float inv_n = n > 0 ? 1.f / static_cast<float>(n) : 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    s_mean[threadIdx.x * V + j] = k[j] + s[j] * inv_n;
    s_m2[threadIdx.x * V + j] = q[j] - s[j] * s[j] * inv_n;
  }
  s_n[threadIdx.x] = n;
  __syncthreads();
  if (t.active && t.rl == 0) {
    // merge the row lanes of this vector in lane order
    float mean[V], m2[V];
    float cnt = static_cast<float>(n);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      mean[j] = s_mean[threadIdx.x * V + j];
      m2[j] = s_m2[threadIdx.x * V + j];
    }
    for (int l = 1; l < t.rpp; ++l) {
      const int o = l * t.cw + t.cl;
      const float nb = static_cast<float>(s_n[o]);
      if (nb == 0.f) continue;
      const float tot = cnt + nb, w = nb / tot;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d = s_mean[o * V + j] - mean[j];
        mean[j] = fmaf(d, w, mean[j]);
        m2[j] += s_m2[o * V + j] + d * d * cnt * w;
      }
      cnt = tot;
    }
    float* out = partial + static_cast<long long>(blockIdx.x) * 2 * C + t.col0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      out[j] = mean[j];
      out[C + j] = m2[j];
    }
  }
}

// The second stage of both reductions: a workgroup takes 16 channels, its 16 x 16 threads split the chunks into 16
// contiguous segments (thread = segment * 16 + channel), every thread folds its segment in index order and the
// segment results are folded in segment order -- a fixed tree, so the result does not depend on timing.
constexpr int kFinChannels = 16, kFinSegments = kBnThreads / kFinChannels;

// Merges the chunks' (count, mean, M2); writes the saved statistics, the affine form of the normalisation and, when
// given, the running statistics (unbiased variance, as torch.nn.BatchNorm2d keeps them).
__global__ __launch_bounds__(kBnThreads) void bn_stats_finalize_kernel(const float* __restrict__ partial, int chunks, long long rows_per_chunk,
                                                                       long long R, long long C, const void* __restrict__ gamma,
                                                                       const void* __restrict__ beta, int param_dtype, float eps, float momentum,
                                                                       void* __restrict__ running_mean, void* __restrict__ running_var,
                                                                       int stat_dtype, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                       float* __restrict__ scale_out, float* __restrict__ shift_out) {
  __shared__ float s_mean[kBnThreads], s_m2[kBnThreads], s_cnt[kBnThreads];
  const int ch = threadIdx.x % kFinChannels, seg = threadIdx.x / kFinChannels;
  const long long c = static_cast<long long>(blockIdx.x) * kFinChannels + ch;
  const int per_seg = (chunks + kFinSegments - 1) / kFinSegments;
  const int k0 = seg * per_seg, k1 = k0 + per_seg < chunks ? k0 + per_seg : chunks;
  float mean = 0.f, m2 = 0.f, cnt = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
      const long long r0 = k * rows_per_chunk;
      const float nb = static_cast<float>((r0 + rows_per_chunk < R ? r0 + rows_per_chunk : R) - r0);
      const float mb = partial[static_cast<long long>(k) * 2 * C + c], qb = partial[static_cast<long long>(k) * 2 * C + C + c];
      const float tot = cnt + nb, w = nb * __builtin_amdgcn_rcpf(tot), d = mb - mean;
      mean = fmaf(d, w, mean);
      m2 += qb + d * d * cnt * w;
      cnt = tot;
    }
  }
  s_mean[threadIdx.x] = mean;
  s_m2[threadIdx.x] = m2;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  if (seg != 0 || c >= C) return;
  for (int sg = 1; sg < kFinSegments; ++sg) {
    const int o = sg * kFinChannels + ch;
    const float nb = s_cnt[o];
    if (nb == 0.f) continue;
    const float tot = cnt + nb, w = nb / tot, d = s_mean[o] - mean;
    mean = fmaf(d, w, mean);
    m2 += s_m2[o] + d * d * cnt * w;
    cnt = tot;
  }
  const float var = fmaxf(m2 / static_cast<float>(R), 0.f);
  const float rstd = rsqrtf(var + eps);
  mean_out[c] = mean;
  rstd_out[c] = rstd;
  const float a = (gamma ? chan_load(gamma, c, param_dtype) : 1.f) * rstd;
  scale_out[c] = a;
  shift_out[c] = (beta ? chan_load(beta, c, param_dtype) : 0.f) - mean * a;
  if (running_mean) {
    const float old = chan_load(running_mean, c, stat_dtype);
    chan_store(running_mean, c, stat_dtype, fmaf(momentum, mean - old, old));
  }
  if (running_var) {
    const float unbiased = R > 1 ? m2 / static_cast<float>(R - 1) : var;
    const float old = chan_load(running_var, c, stat_dtype);
    chan_store(running_var, c, stat_dtype, fmaf(momentum, unbiased - old, old));
  }
}

// y = x * a[c] + b[c] (+ residual), optionally through a ReLU.  Every thread keeps its channel vector for the whole
// launch (the thread count is a multiple of the vectors per row), so a and b are loaded once.
template <typename TR, bool RELU, bool ADD>
__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const typename TR::T* __restrict__ x, const typename TR::T* __restrict__ res,
                                                              typename TR::T* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, long long n_vec, long long cv) {
  constexpr int V = TR::kVec;
  const long long stride = static_cast<long long>(gridDim.x) * kBnThreads;
  long long i = static_cast<long long>(blockIdx.x) * kBnThreads + threadIdx.x;
  if (i >= n_vec) return;
  const long long c0 = (i % cv) * V;
  float a[V], b[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    a[j] = scale[c0 + j];
    b[j] = shift[c0 + j];
  }
  for (; i < n_vec; i += stride) {
    float v[V];
    load_as_f32<TR, V>(x + i * V, v);
    if (ADD) {
      float r[V];
      load_as_f32<TR, V>(res + i * V, r);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = fmaf(v[j], a[j], b[j]) + r[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = fmaf(v[j], a[j], b[j]);
    }
    if (RELU) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    store_from_f32<TR, V>(y + i * V, v);
  }
}

// How the backward finds the ReLU mask: not at all, from the saved output (y > 0), or -- when nothing was added
// before the ReLU -- from x itself: y = relu(x * a + b) is positive exactly where x * a + b is (rounding the fp32
// value to the activation dtype cannot change its sign), with a | b the forward's own coefficients, so the output
// need not be read.
constexpr int kMaskNone = 0, kMaskFromY = 1, kMaskFromX = 2;

// partial[chunk][0][c] = sum g, partial[chunk][1][c] = sum g * xhat over the chunk's rows; g = dy under the ReLU mask.
template <typename TR, int MASK>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_partial_kernel(const typename TR::T* __restrict__ dy, const typename TR::T* __restrict__ x,
                                                                    const typename TR::T* __restrict__ y, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ scale_shift,
                                                                    long long R, long long C, long long rows_per_chunk,
                                                                    float* __restrict__ partial) {
  constexpr int V = TR::kVec;
  __shared__ float s_a[kBnThreads * V], s_b[kBnThreads * V];
  const BnTile t = bn_tile<V>(C);
  const long long r0 = static_cast<long long>(blockIdx.x) * rows_per_chunk;
  const long long r1 = r0 + rows_per_chunk < R ? r0 + rows_per_chunk : R;
  float s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.f;
  if (t.active) {
    float mu[V], rs[V], a[V], b[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      mu[j] = mean[t.col0 + j];
      rs[j] = rstd[t.col0 + j];
      a[j] = MASK == kMaskFromX ? scale_shift[t.col0 + j] : 0.f;
      b[j] = MASK == kMaskFromX ? scale_shift[C + t.col0 + j] : 0.f;
    }
    for (long long r = r0 + t.rl; r < r1; r += t.rpp) {
      float g[V], xv[V];
      load_as_f32<TR, V>(dy + r * C + t.col0, g);
      load_as_f32<TR, V>(x + r * C + t.col0, xv);
      if (MASK == kMaskFromY) {
        float yv[V];
        load_as_f32<TR, V>(y + r * C + t.col0, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
      } else if (MASK == kMaskFromX) {
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = fmaf(xv[j], a[j], b[j]) > 0.f ? g[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s1[j] += g[j];
        s2[j] = fmaf(g[j], (xv[j] - mu[j]) * rs[j], s2[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    s_a[threadIdx.x * V + j] = s1[j];
    s_b[threadIdx.x * V + j] = s2[j];
  }
  __syncthreads();
  if (t.active && t.rl == 0) {
    for (int l = 1; l < t.rpp; ++l) {
      const int o = l * t.cw + t.cl;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s1[j] += s_a[o * V + j];
        s2[j] += s_b[o * V + j];
      }
    }
    float* out = partial + static_cast<long long>(blockIdx.x) * 2 * C + t.col0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      out[j] = s1[j];
      out[C + j] = s2[j];
    }
  }
}

// dbeta = sum g, dgamma = sum g * xhat; coef[0][c] = gamma * rstd, coef[1][c] = dbeta / R, coef[2][c] = dgamma / R.
__global__ __launch_bounds__(kBnThreads) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, long long R, long long C,
                                                                     const void* __restrict__ gamma, int param_dtype,
                                                                     const float* __restrict__ rstd, void* __restrict__ dgamma,
                                                                     void* __restrict__ dbeta, float* __restrict__ coef) {
  __shared__ float s_1[kBnThreads], s_2[kBnThreads];
  const int ch = threadIdx.x % kFinChannels, seg = threadIdx.x / kFinChannels;
  const long long c = static_cast<long long>(blockIdx.x) * kFinChannels + ch;
  const int per_seg = (chunks + kFinSegments - 1) / kFinSegments;
  const int k0 = seg * per_seg, k1 = k0 + per_seg < chunks ? k0 + per_seg : chunks;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
      s1 += partial[static_cast<long long>(k) * 2 * C + c];
      s2 += partial[static_cast<long long>(k) * 2 * C + C + c];
    }
  }
  s_1[threadIdx.x] = s1;
  s_2[threadIdx.x] = s2;
  __syncthreads();
  if (seg != 0 || c >= C) return;
  for (int sg = 1; sg < kFinSegments; ++sg) {
    s1 += s_1[sg * kFinChannels + ch];
    s2 += s_2[sg * kFinChannels + ch];
  }
  if (dbeta) chan_store(dbeta, c, param_dtype, s1);
  if (dgamma) chan_store(dgamma, c, param_dtype, s2);
  const float inv = 1.f / static_cast<float>(R);
  coef[c] = (gamma ? chan_load(gamma, c, param_dtype) : 1.f) * rstd[c];
  coef[C + c] = s1 * inv;
  coef[2 * C + c] = s2 * inv;
}

// dx = (g - c1 - xhat * c2) * a; dres = g (when the block's residual branch needs its gradient).
template <typename TR, int MASK, bool DRES>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_apply_kernel(const typename TR::T* __restrict__ dy, const typename TR::T* __restrict__ x,
                                                                  const typename TR::T* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ coef,
                                                                  const float* __restrict__ scale_shift, typename TR::T* __restrict__ dx,
                                                                  typename TR::T* __restrict__ dres, long long n_vec, long long cv, long long C) {
  constexpr int V = TR::kVec;
  const long long stride = static_cast<long long>(gridDim.x) * kBnThreads;
  long long i = static_cast<long long>(blockIdx.x) * kBnThreads + threadIdx.x;
  if (i >= n_vec) return;
  const long long c0 = (i % cv) * V;
  float mu[V], rs[V], a[V], c1[V], c2[V], fa[V], fb[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    mu[j] = mean[c0 + j];
    rs[j] = rstd[c0 + j];
    a[j] = coef[c0 + j];
    c1[j] = coef[C + c0 + j];
    c2[j] = coef[2 * C + c0 + j];
    fa[j] = MASK == kMaskFromX ? scale_shift[c0 + j] : 0.f;
    fb[j] = MASK == kMaskFromX ? scale_shift[C + c0 + j] : 0.f;
  }
  for (; i < n_vec; i += stride) {
    float g[V], xv[V];
    load_as_f32<TR, V>(dy + i * V, g);
    load_as_f32<TR, V>(x + i * V, xv);
    if (MASK == kMaskFromY) {
      float yv[V];
      load_as_f32<TR, V>(y + i * V, yv);
#pragma unroll
      for (int j = 0; j < V; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
    } else if (MASK == kMaskFromX) {
#pragma unroll
      for (int j = 0; j < V; ++j) g[j] = fmaf(xv[j], fa[j], fb[j]) > 0.f ? g[j] : 0.f;
    }
    if (DRES) store_from_f32<TR, V>(dres + i * V, g);
    float d[V];
#pragma unroll
    for (int j = 0; j < V; ++j) d[j] = (g[j] - c1[j] - (xv[j] - mu[j]) * rs[j] * c2[j]) * a[j];
    store_from_f32<TR, V>(dx + i * V, d);
  }
}

// Row chunks: enough workgroups to fill the chip a few times over, at least 32 passes of rows each.
void bn_chunks(long long R, long long C, int vec, long long& rows_per_chunk, int& chunks, int& col_blocks) {
  const long long cv = C / vec;
  col_blocks = static_cast<int>((cv + kBnThreads - 1) / kBnThreads);
  const long long cw = cv < kBnThreads ? cv : kBnThreads;
  const long long rpp = kBnThreads / cw;
  long long want = 1024 / col_blocks;
  if (want < 1) want = 1;
  rows_per_chunk = (R + want - 1) / want;
  const long long min_rows = 32 * rpp;
  if (rows_per_chunk < min_rows) rows_per_chunk = min_rows;
  rows_per_chunk = (rows_per_chunk + rpp - 1) / rpp * rpp;
  chunks = static_cast<int>((R + rows_per_chunk - 1) / rows_per_chunk);
}

unsigned bn_stream_grid(long long n_vec, long long cv) {
  // a multiple of 8 workgroups, so that the thread count is a multiple of any vectors-per-row up to 2048
  long long g = (n_vec + kBnThreads - 1) / kBnThreads;
  const long long cap = 256 * 16;
  if (g > cap) g = cap;
  g = (g + 7) / 8 * 8;
  (void)cv;
  return static_cast<unsigned>(g);
}

}  // namespace

bool batchnorm_supported(long long R, long long C, int dtype) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (R < 1 || C < vec || C % vec != 0) return false;
  const long long cv = C / vec;
  // the streaming kernels keep a thread on one channel vector: the thread count (a multiple of 2048) must be a multiple of cv;
  // the reduction kernels split 256 threads into whole rows of vectors
  if ((8 * kBnThreads) % cv != 0) return false;
  const long long cw = cv < kBnThreads ? cv : kBnThreads;
  return kBnThreads % cw == 0 && cv % cw == 0;
}

long long batchnorm_workspace(long long R, long long C, int dtype) {
  long long rows_per_chunk;
  int chunks, col_blocks;
  bn_chunks(R, C, dtype == kF32 ? 4 : 8, rows_per_chunk, chunks, col_blocks);
  return static_cast<long long>(chunks) * 2 * C + 3 * C;  // partials, then the three backward coefficients
}

void launch_batchnorm_fwd(const void* x, const void* res, void* y, const void* gamma, const void* beta, int param_dtype, void* running_mean,
                          void* running_var, int stat_dtype, float momentum, float eps, float* mean, float* rstd, float* scale_shift,
                          float* workspace, long long R, long long C, int relu, int dtype, hipStream_t stream) {
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    constexpr int V = TR::kVec;
    long long rows_per_chunk;
    int chunks, col_blocks;
    bn_chunks(R, C, V, rows_per_chunk, chunks, col_blocks);
    hipLaunchKernelGGL((bn_stats_partial_kernel<TR>), dim3(chunks, col_blocks), dim3(kBnThreads), 0, stream, static_cast<const T*>(x), R, C,
                       rows_per_chunk, workspace);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(static_cast<unsigned>((C + kFinChannels - 1) / kFinChannels)), dim3(kBnThreads), 0, stream,
                       workspace, chunks, rows_per_chunk, R, C, gamma, beta, param_dtype, eps, momentum, running_mean, running_var, stat_dtype,
                       mean, rstd, scale_shift, scale_shift + C);
    const long long cv = C / V, n_vec = R * cv;
    const dim3 grid(bn_stream_grid(n_vec, cv));
    const T* xp = static_cast<const T*>(x);
    const T* rp = static_cast<const T*>(res);
    T* yp = static_cast<T*>(y);
    auto go = [&](auto relu_c, auto add_c) {
      hipLaunchKernelGGL((bn_apply_kernel<TR, decltype(relu_c)::value, decltype(add_c)::value>), grid, dim3(kBnThreads), 0, stream, xp, rp, yp,
                         scale_shift, scale_shift + C, n_vec, cv);
    };
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    if (relu && res) go(yes, yes);
    else if (relu) go(yes, no);
    else if (res) go(no, yes);
    else go(no, no);
  });
}

void launch_batchnorm_apply(const void* x, const void* res, void* y, const float* scale, const float* shift, long long R, long long C, int relu,
                            int dtype, hipStream_t stream) {
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    constexpr int V = TR::kVec;
    const long long cv = C / V, n_vec = R * cv;
    const dim3 grid(bn_stream_grid(n_vec, cv));
    const T* xp = static_cast<const T*>(x);
    const T* rp = static_cast<const T*>(res);
    T* yp = static_cast<T*>(y);
    auto go = [&](auto relu_c, auto add_c) {
      hipLaunchKernelGGL((bn_apply_kernel<TR, decltype(relu_c)::value, decltype(add_c)::value>), grid, dim3(kBnThreads), 0, stream, xp, rp, yp,
                         scale, shift, n_vec, cv);
    };
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    if (relu && res) go(yes, yes);
    else if (relu) go(yes, no);
    else if (res) go(no, yes);
    else go(no, no);
  });
}

void launch_batchnorm_bwd(const void* dy, const void* x, const void* y, const float* scale_shift, const void* gamma, int param_dtype,
                          const float* mean, const float* rstd, void* dx, void* dres, void* dgamma, void* dbeta, float* workspace, long long R,
                          long long C, int relu, int dtype, hipStream_t stream) {
  // the mask comes from x when the forward's coefficients are given (no residual was added), else from y
  const int mask = !relu ? kMaskNone : (scale_shift != nullptr ? kMaskFromX : kMaskFromY);
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    constexpr int V = TR::kVec;
    long long rows_per_chunk;
    int chunks, col_blocks;
    bn_chunks(R, C, V, rows_per_chunk, chunks, col_blocks);
    float* coef = workspace + static_cast<long long>(chunks) * 2 * C;
    const T* dyp = static_cast<const T*>(dy);
    const T* xp = static_cast<const T*>(x);
    const T* yp = static_cast<const T*>(y);
    T* dxp = static_cast<T*>(dx);
    T* drp = static_cast<T*>(dres);
    const long long cv = C / V, n_vec = R * cv;
    const dim3 grid(bn_stream_grid(n_vec, cv));
    auto go = [&](auto mask_c) {
      constexpr int M = decltype(mask_c)::value;
      hipLaunchKernelGGL((bn_bwd_partial_kernel<TR, M>), dim3(chunks, col_blocks), dim3(kBnThreads), 0, stream, dyp, xp, yp, mean, rstd,
                         scale_shift, R, C, rows_per_chunk, workspace);
      hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(static_cast<unsigned>((C + kFinChannels - 1) / kFinChannels)), dim3(kBnThreads), 0, stream,
                         workspace, chunks, R, C, gamma, param_dtype, rstd, dgamma, dbeta, coef);
      if (dres)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<TR, M, true>), grid, dim3(kBnThreads), 0, stream, dyp, xp, yp, mean, rstd, coef, scale_shift, dxp,
                           drp, n_vec, cv, C);
      else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<TR, M, false>), grid, dim3(kBnThreads), 0, stream, dyp, xp, yp, mean, rstd, coef, scale_shift, dxp,
                           drp, n_vec, cv, C);
    };
    if (mask == kMaskFromX) go(std::integral_constant<int, kMaskFromX>{});
    else if (mask == kMaskFromY) go(std::integral_constant<int, kMaskFromY>{});
    else go(std::integral_constant<int, kMaskNone>{});
  });
}

}  // namespace kernels
}  // namespace dpa
