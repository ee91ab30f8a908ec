// Layer normalisation forward / backward for gfx950 (used by the GPT-2 configuration).
//
// One 64-lane wave owns one row and keeps it in registers: the row is read from HBM once,
// mean and variance are two wave-shuffle reductions, and the result is written once
// (16-byte accesses per lane).  The residual add in front of a transformer's layer norm is folded
// in: with `res` the kernel normalises s = x + res and also writes s (rounded to the storage type
// first, so the result equals the unfused add followed by layer norm), and the backward adds the
// gradient arriving at s from the residual path (`dres`) to the layer-norm gradient -- the separate
// elementwise add passes disappear in both directions.  The backward pass does the same per row for dx, and because
// a lane always owns the same columns it accumulates its share of dgamma / dbeta in
// registers across all rows its workgroup visits; workgroups then emit one partial row
// each and a second small kernel sums them (deterministic, no float atomics).
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {
#ifndef LN_STREAM_STORES
#define LN_STREAM_STORES 1
#endif
#ifndef LN_STREAM_LOADS
#define LN_STREAM_LOADS 1
#endif
constexpr bool kStream = LN_STREAM_STORES != 0, kStreamLoads = LN_STREAM_LOADS != 0;


constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxBlocks = 2048;

// V = elements per lane and access (16 or 8 bytes), MAXV = accesses per lane and row; a row of
// `cols` needs cols <= 64 * V * MAXV.  The host picks the 8-byte form when it divides the row evenly
// over the 64 lanes and the 16-byte form does not (768 columns of bf16: 3 x 4 elements on every lane
// instead of 2 x 8 on half of them), which saves a quarter of the registers and of the arithmetic.
// Both kernels are bound by memory-level parallelism, not arithmetic: a wave needs its whole row
// before the first reduction, so each wave requests the NEXT row it will visit (kept packed in
// registers) before it starts on the current one, and the per-column parameters are read once per
// wave instead of once per row.
template <typename TR, int V, int MAXV>
__global__ __launch_bounds__(kThreads) void layernorm_fwd_kernel(const typename TR::T* __restrict__ x,
                                                                 const typename TR::T* __restrict__ res,
                                                                 const typename TR::T* __restrict__ res_bias,
                                                                 typename TR::T* __restrict__ sum_out,
                                                                 const typename TR::T* __restrict__ gamma,
                                                                 const typename TR::T* __restrict__ beta,
                                                                 typename TR::T* __restrict__ y, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out, long long rows, int cols,
                                                                 float eps) {
  using Raw = raw_t<TR, V>;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nvec = cols / V;
  const float inv_n = 1.f / static_cast<float>(cols);
  const long long stride = static_cast<long long>(gridDim.x) * kWaves;
  long long row = static_cast<long long>(blockIdx.x) * kWaves + wid;

  Raw gam[MAXV], bet[MAXV], rb[MAXV], nx[MAXV], nr[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + 64 * i;
    gam[i] = bet[i] = rb[i] = nx[i] = nr[i] = zero_raw<TR, V>();
    if (vi < nvec) {
      gam[i] = load_raw<TR, V>(gamma + vi * V);
      bet[i] = load_raw<TR, V>(beta + vi * V);
      if (res_bias) rb[i] = load_raw<TR, V>(res_bias + vi * V);
    }
  }
  auto request = [&](long long r) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        nx[i] = kStreamLoads ? load_raw_stream<TR, V>(x + r * cols + vi * V) : load_raw<TR, V>(x + r * cols + vi * V);
        if (res) nr[i] = kStreamLoads ? load_raw_stream<TR, V>(res + r * cols + vi * V) : load_raw<TR, V>(res + r * cols + vi * V);
      }
    }
  };
  if (row < rows) request(row);
  for (; row < rows; row += stride) {
    Raw cx[MAXV], cr[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      cx[i] = nx[i];
      cr[i] = nr[i];
    }
    if (row + stride < rows) request(row + stride);
    float v[MAXV][V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      keep_packed(gam[i]);
      keep_packed(bet[i]);
      keep_packed(rb[i]);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        raw_to_f32<TR, V>(cx[i], v[i]);
        if (res) {
          float r[V];
          raw_to_f32<TR, V>(cr[i], r);
          if (res_bias) {  // the bias of the layer that produced `res`, added here instead of in its GEMM
            float b[V];
            raw_to_f32<TR, V>(rb[i], b);
#pragma unroll
            for (int j = 0; j < V; ++j) r[j] = TR::to_f32(TR::from_f32(r[j] + b[j]));
          }
#pragma unroll
          for (int j = 0; j < V; ++j) v[i][j] = TR::to_f32(TR::from_f32(v[i][j] + r[j]));
          store_raw_from_f32<TR, V, kStream>(sum_out + row * cols + vi * V, v[i]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) s += v[i][j];
      }
    }
    const float mean = wave_sum(s) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + 64 * i < nvec) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float d = v[i][j] - mean;
          sq += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_n + eps);
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    typename TR::T* yr = y + row * cols;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        float g[V], b[V], o[V];
        raw_to_f32<TR, V>(gam[i], g);
        raw_to_f32<TR, V>(bet[i], b);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
        store_raw_from_f32<TR, V, kStream>(yr + vi * V, o);
      }
    }
  }
}

// DSUM: also accumulate the column sums of dx (the gradient of a bias that was added to the
// residual operand in the forward); the partial rows then hold 3*cols values instead of 2*cols.
template <typename TR, int V, int MAXV, bool DSUM>
__global__ __launch_bounds__(kThreads, V* MAXV <= 12 ? 4 : (V * MAXV <= 16 ? 3 : 1)) void layernorm_bwd_kernel(
    const typename TR::T* __restrict__ dy, const typename TR::T* __restrict__ x, const typename TR::T* __restrict__ dres,
    const typename TR::T* __restrict__ gamma, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    typename TR::T* __restrict__ dx, float* __restrict__ partial, long long rows, int cols) {
  using Raw = raw_t<TR, V>;
  extern __shared__ __attribute__((aligned(16))) float ln_smem[];  // [NACC][cols]
  constexpr int NACC = DSUM ? 3 : 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nvec = cols / V;
  const float inv_n = 1.f / static_cast<float>(cols);
  const long long stride = static_cast<long long>(gridDim.x) * kWaves;
  long long row = static_cast<long long>(blockIdx.x) * kWaves + wid;

  Raw gam[MAXV], nx[MAXV], ng[MAXV];
  float dg[MAXV][V], db[MAXV][V], ds[DSUM ? MAXV : 1][V];
  float nmean = 0.f, nrstd = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + 64 * i;
    gam[i] = nx[i] = ng[i] = zero_raw<TR, V>();
    if (vi < nvec) gam[i] = load_raw<TR, V>(gamma + vi * V);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      dg[i][j] = db[i][j] = 0.f;
      if (DSUM) ds[i][j] = 0.f;
    }
  }
  auto request = [&](long long r) __attribute__((always_inline)) {
    nmean = mean_in[r];
    nrstd = rstd_in[r];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        nx[i] = kStreamLoads ? load_raw_stream<TR, V>(x + r * cols + vi * V) : load_raw<TR, V>(x + r * cols + vi * V);
        ng[i] = kStreamLoads ? load_raw_stream<TR, V>(dy + r * cols + vi * V) : load_raw<TR, V>(dy + r * cols + vi * V);
      }
    }
  };
  if (row < rows) request(row);
  for (; row < rows; row += stride) {
    const float mean = nmean, rstd = nrstd;
    Raw cx[MAXV], cg[MAXV], cr[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      cx[i] = nx[i];
      cg[i] = ng[i];
      cr[i] = zero_raw<TR, V>();
    }
    if (row + stride < rows) request(row + stride);
    // the residual-path gradient is only needed after the reductions: requested now, for this row
    if (dres) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nvec) cr[i] = kStreamLoads ? load_raw_stream<TR, V>(dres + row * cols + (lane + 64 * i) * V) : load_raw<TR, V>(dres + row * cols + (lane + 64 * i) * V);
    }
    // The row stays packed across the two reductions and is expanded a second time afterwards:
    // 3 VALU operations per element are cheaper than the registers fp32 copies would hold.
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) keep_packed(gam[i]);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        float g[V], xh[V], gy[V];
        raw_to_f32<TR, V>(cx[i], xh);
        raw_to_f32<TR, V>(cg[i], gy);
        raw_to_f32<TR, V>(gam[i], g);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          xh[j] = (xh[j] - mean) * rstd;
          dg[i][j] += gy[j] * xh[j];
          db[i][j] += gy[j];
          gy[j] *= g[j];
          s1 += gy[j];
          s2 += gy[j] * xh[j];
        }
      }
    }
    s1 = wave_sum(s1) * inv_n;
    s2 = wave_sum(s2) * inv_n;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      keep_packed(cx[i]);
      keep_packed(cg[i]);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        float g[V], xh[V], gy[V], o[V];
        raw_to_f32<TR, V>(cx[i], xh);
        raw_to_f32<TR, V>(cg[i], gy);
        raw_to_f32<TR, V>(gam[i], g);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = rstd * (gy[j] * g[j] - s1 - (xh[j] - mean) * rstd * s2);
        if (dres) {
          float r[V];
          raw_to_f32<TR, V>(cr[i], r);
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] += r[j];
        }
        if (DSUM) {
#pragma unroll
          for (int j = 0; j < V; ++j) ds[i][j] += o[j];
        }
        store_raw_from_f32<TR, V, kStream>(dx + row * cols + vi * V, o);
      }
    }
  }

  // Combine the workgroup's waves through one LDS row set, wave after wave (a lane owns the same
  // columns in every wave, so the turns need no atomics), then one partial row per workgroup.
  for (int w = 0; w < kWaves; ++w) {
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const int c = vi * V + j;
            if (w == 0) {
              ln_smem[c] = dg[i][j];
              ln_smem[cols + c] = db[i][j];
              if (DSUM) ln_smem[2 * cols + c] = ds[i][j];
            } else {
              ln_smem[c] += dg[i][j];
              ln_smem[cols + c] += db[i][j];
              if (DSUM) ln_smem[2 * cols + c] += ds[i][j];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  float* out = partial + static_cast<size_t>(blockIdx.x) * NACC * cols;
  for (int c = threadIdx.x; c < NACC * cols; c += kThreads) out[c] = ln_smem[c];
}

// dgamma / dbeta = column sums of the per-workgroup partial rows.  16 outputs per workgroup, 16 row
// groups x 16 columns of threads (64-byte row segments), four independent loads in flight per
// thread (the kernel is latency-bound: a thread walks nblocks/16 rows), LDS combine.
template <typename TR>
__global__ __launch_bounds__(kThreads) void layernorm_bwd_reduce_kernel(const float* __restrict__ partial, typename TR::T* __restrict__ dgamma,
                                                                        typename TR::T* __restrict__ dbeta, typename TR::T* __restrict__ dsum,
                                                                        int cols, int nblocks, int nacc) {
  __shared__ float red[16][17];
  const int cc = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cc;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < nacc * cols) {
    const size_t ld = static_cast<size_t>(nacc) * cols;
    int b = rg;
    for (; b + 48 < nblocks; b += 64) {
      s0 += partial[static_cast<size_t>(b) * ld + c];
      s1 += partial[static_cast<size_t>(b + 16) * ld + c];
      s2 += partial[static_cast<size_t>(b + 32) * ld + c];
      s3 += partial[static_cast<size_t>(b + 48) * ld + c];
    }
    for (; b < nblocks; b += 16) s0 += partial[static_cast<size_t>(b) * ld + c];
  }
  red[rg][cc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < nacc * cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cc];
    if (c < cols) dgamma[c] = TR::from_f32(t);
    else if (c < 2 * cols) dbeta[c - cols] = TR::from_f32(t);
    else dsum[c - 2 * cols] = TR::from_f32(t);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Rows a wave cannot keep in registers (more than 64 * 16 bytes * 8 (forward) / * 4 (backward)), or whose width is
// not a multiple of the 16-byte vector: one workgroup per row, element-wise coalesced accesses, the row read again
// from the caches for the second and third pass.  Same arithmetic, in the same order per element, as the kernels
// above; dgamma / dbeta accumulate in the registers of the thread that owns the column (kRowCols per thread) and
// leave as one partial row per workgroup for layernorm_bwd_reduce_kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int kRowCols = 32;                               // columns per thread the backward can accumulate
constexpr long long kRowMaxCols = kRowCols * kThreads;     // 8192

template <typename TR>
__global__ __launch_bounds__(kThreads) void layernorm_fwd_row_kernel(const typename TR::T* __restrict__ x, const typename TR::T* __restrict__ res,
                                                                     const typename TR::T* __restrict__ res_bias,
                                                                     typename TR::T* __restrict__ sum_out, const typename TR::T* __restrict__ gamma,
                                                                     const typename TR::T* __restrict__ beta, typename TR::T* __restrict__ y,
                                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, long long rows,
                                                                     int cols, float eps) {
  __shared__ float scratch[kWaves];
  const float inv_n = 1.f / static_cast<float>(cols);
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const typename TR::T* xr = x + row * cols;
    const typename TR::T* rr = res ? res + row * cols : nullptr;
    typename TR::T* sr = res ? sum_out + row * cols : nullptr;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += kThreads) {
      float v = TR::to_f32(xr[c]);
      if (res) {
        float r = TR::to_f32(rr[c]);
        if (res_bias) r = TR::to_f32(TR::from_f32(r + TR::to_f32(res_bias[c])));
        const typename TR::T sv = TR::from_f32(v + r);
        sr[c] = sv;  // read back below by the same thread
        v = TR::to_f32(sv);
      }
      s += v;
    }
    const float mean = block_sum<kWaves>(s, scratch) * inv_n;
    const typename TR::T* vr = res ? sr : xr;
    float sq = 0.f;
    for (int c = threadIdx.x; c < cols; c += kThreads) {
      const float d = TR::to_f32(vr[c]) - mean;
      sq += d * d;
    }
    const float rstd = rsqrtf(block_sum<kWaves>(sq, scratch) * inv_n + eps);
    if (threadIdx.x == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    typename TR::T* yr = y + row * cols;
    for (int c = threadIdx.x; c < cols; c += kThreads)
      yr[c] = TR::from_f32((TR::to_f32(vr[c]) - mean) * rstd * TR::to_f32(gamma[c]) + TR::to_f32(beta[c]));
  }
}

template <typename TR, bool DSUM>
__global__ __launch_bounds__(kThreads) void layernorm_bwd_row_kernel(const typename TR::T* __restrict__ dy, const typename TR::T* __restrict__ x,
                                                                     const typename TR::T* __restrict__ dres,
                                                                     const typename TR::T* __restrict__ gamma, const float* __restrict__ mean_in,
                                                                     const float* __restrict__ rstd_in, typename TR::T* __restrict__ dx,
                                                                     float* __restrict__ partial, long long rows, int cols) {
  __shared__ float scratch[kWaves];
  constexpr int NACC = DSUM ? 3 : 2;
  const float inv_n = 1.f / static_cast<float>(cols);
  float dg[kRowCols], db[kRowCols], ds[DSUM ? kRowCols : 1];
#pragma unroll
  for (int k = 0; k < kRowCols; ++k) {
    dg[k] = db[k] = 0.f;
    if (DSUM) ds[k] = 0.f;
  }
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    const typename TR::T* gr = dy + row * cols;
    const typename TR::T* xr = x + row * cols;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kRowCols; ++k) {
      const int c = threadIdx.x + k * kThreads;
      if (c < cols) {
        const float xh = (TR::to_f32(xr[c]) - mean) * rstd;
        float gy = TR::to_f32(gr[c]);
        dg[k] += gy * xh;
        db[k] += gy;
        gy *= TR::to_f32(gamma[c]);
        s1 += gy;
        s2 += gy * xh;
      }
    }
    s1 = block_sum<kWaves>(s1, scratch) * inv_n;
    s2 = block_sum<kWaves>(s2, scratch) * inv_n;
#pragma unroll
    for (int k = 0; k < kRowCols; ++k) {
      const int c = threadIdx.x + k * kThreads;
      if (c < cols) {
        float o = rstd * (TR::to_f32(gr[c]) * TR::to_f32(gamma[c]) - s1 - (TR::to_f32(xr[c]) - mean) * rstd * s2);
        if (dres) o += TR::to_f32(dres[row * cols + c]);
        if (DSUM) ds[k] += o;
        dx[row * cols + c] = TR::from_f32(o);
      }
    }
  }
  float* out = partial + static_cast<size_t>(blockIdx.x) * NACC * cols;
#pragma unroll
  for (int k = 0; k < kRowCols; ++k) {
    const int c = threadIdx.x + k * kThreads;
    if (c < cols) {
      out[c] = dg[k];
      out[cols + c] = db[k];
      if (DSUM) out[2 * cols + c] = ds[k];
    }
  }
}

int pick_blocks(long long rows) {
  long long b = (rows + kWaves - 1) / kWaves;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

// Calls f(V, MAXV) as integral constants for a row of `cols` elements: the 8-byte access form when it
// spreads the row evenly over the wave and the 16-byte form does not, the 16-byte form otherwise.
template <typename TR, typename F>
bool dispatch_shape(long long cols, int limit16, F&& f) {
  constexpr int V16 = TR::kVec, V8 = TR::kVec / 2;
  const long long n16 = cols / V16, n8 = cols / V8;
  if (n16 % 64 != 0 && cols % V8 == 0 && n8 % 64 == 0) {
    const long long m = n8 / 64;
    if (m == 1) { f(std::integral_constant<int, V8>{}, std::integral_constant<int, 1>{}); return true; }
    if (m == 3) { f(std::integral_constant<int, V8>{}, std::integral_constant<int, 3>{}); return true; }
    if (m == 5) { f(std::integral_constant<int, V8>{}, std::integral_constant<int, 5>{}); return true; }
  }
  const int need = static_cast<int>((n16 + 63) / 64);
  if (need > limit16) return false;
  if (need <= 1) f(std::integral_constant<int, V16>{}, std::integral_constant<int, 1>{});
  else if (need <= 2) f(std::integral_constant<int, V16>{}, std::integral_constant<int, 2>{});
  else if (need <= 4) f(std::integral_constant<int, V16>{}, std::integral_constant<int, 4>{});
  else f(std::integral_constant<int, V16>{}, std::integral_constant<int, 8>{});
  return true;
}

}  // namespace

// The wave-per-row backward kernel: rows of a whole number of 16-byte vectors, at most 4 per lane, and at most 1365
// columns when the column sums of dx are wanted as well (its three accumulator rows live in LDS).
bool layernorm_bwd_in_registers(long long cols, int dtype, bool want_dsum) {
  const long long vec = dtype == kF32 ? 4 : 8;
  return cols % vec == 0 && cols <= 64 * vec * 4 && (!want_dsum || cols <= 1365);
}
bool layernorm_supported(long long cols) { return cols >= 1 && cols <= kRowMaxCols; }

void launch_layernorm_fwd(const void* x, const void* res, const void* res_bias, void* sum_out, const void* gamma, const void* beta, void* y,
                          float* mean, float* rstd, long long rows, long long cols, float eps, int dtype, hipStream_t stream) {
  if (rows <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    long long blocks = (rows + kWaves - 1) / kWaves;
    if (blocks > 4096) blocks = 4096;
    const bool in_registers = cols % TR::kVec == 0 && dispatch_shape<TR>(cols, 8, [&](auto vv, auto mv) {
      constexpr int V = decltype(vv)::value, MAXV = decltype(mv)::value;
      hipLaunchKernelGGL((layernorm_fwd_kernel<TR, V, MAXV>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream,
                         static_cast<const T*>(x), static_cast<const T*>(res), static_cast<const T*>(res_bias), static_cast<T*>(sum_out),
                         static_cast<const T*>(gamma),
                         static_cast<const T*>(beta), static_cast<T*>(y), mean, rstd, rows, static_cast<int>(cols), eps);
    });
    if (!in_registers) {
      const long long row_blocks = rows < 8192 ? rows : 8192;
      hipLaunchKernelGGL((layernorm_fwd_row_kernel<TR>), dim3(static_cast<unsigned>(row_blocks)), dim3(kThreads), 0, stream,
                         static_cast<const T*>(x), static_cast<const T*>(res), static_cast<const T*>(res_bias), static_cast<T*>(sum_out),
                         static_cast<const T*>(gamma), static_cast<const T*>(beta), static_cast<T*>(y), mean, rstd, rows, static_cast<int>(cols),
                         eps);
    }
  });
}

long long layernorm_bwd_workspace(long long rows, long long cols) { return 3 * cols * pick_blocks(rows); }

void launch_layernorm_bwd(const void* dy, const void* x, const void* dres, const void* gamma, const float* mean, const float* rstd, void* dx,
                          void* dgamma, void* dbeta, void* dsum, float* workspace, long long rows, long long cols, int dtype,
                          hipStream_t stream) {
  if (rows <= 0) return;
  dispatch_dtype(dtype, [&](auto tr) {
    using TR = decltype(tr);
    using T = typename TR::T;
    const int blocks = pick_blocks(rows);
    const int nacc = dsum ? 3 : 2;
    const size_t lds = static_cast<size_t>(nacc) * cols * sizeof(float);
    const bool in_registers = layernorm_bwd_in_registers(cols, dtype, dsum != nullptr);
    if (!in_registers) {
      if (dsum)
        hipLaunchKernelGGL((layernorm_bwd_row_kernel<TR, true>), dim3(blocks), dim3(kThreads), 0, stream, static_cast<const T*>(dy),
                           static_cast<const T*>(x), static_cast<const T*>(dres), static_cast<const T*>(gamma), mean, rstd, static_cast<T*>(dx),
                           workspace, rows, static_cast<int>(cols));
      else
        hipLaunchKernelGGL((layernorm_bwd_row_kernel<TR, false>), dim3(blocks), dim3(kThreads), 0, stream, static_cast<const T*>(dy),
                           static_cast<const T*>(x), static_cast<const T*>(dres), static_cast<const T*>(gamma), mean, rstd, static_cast<T*>(dx),
                           workspace, rows, static_cast<int>(cols));
    }
    if (in_registers) dispatch_shape<TR>(cols, 4, [&](auto vv, auto mv) {
      constexpr int V = decltype(vv)::value;
      constexpr int MAXV = decltype(mv)::value > 5 ? 4 : decltype(mv)::value;
      auto go = [&](auto ds_tag) {
        constexpr bool DS = decltype(ds_tag)::value;
        hipLaunchKernelGGL((layernorm_bwd_kernel<TR, V, MAXV, DS>), dim3(blocks), dim3(kThreads), lds, stream, static_cast<const T*>(dy),
                           static_cast<const T*>(x), static_cast<const T*>(dres), static_cast<const T*>(gamma), mean, rstd, static_cast<T*>(dx),
                           workspace, rows, static_cast<int>(cols));
      };
      if (dsum) go(std::true_type{});
      else go(std::false_type{});
    });
    hipLaunchKernelGGL((layernorm_bwd_reduce_kernel<TR>), dim3(static_cast<unsigned>((nacc * cols + 15) / 16)), dim3(kThreads), 0, stream,
                       workspace, static_cast<T*>(dgamma), static_cast<T*>(dbeta), static_cast<T*>(dsum), static_cast<int>(cols), blocks, nacc);
  });
}

}  // namespace kernels
}  // namespace dpa
