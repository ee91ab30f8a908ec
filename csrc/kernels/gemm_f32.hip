// fp32 matrix products on the gfx950 matrix cores, with fused bias / activation epilogue.
//
// Provides the nn.Linear forward and backward products of the reference model
// (reference min_DDP.py:41-49, 98, 103) for GEMM-shaped layers in full fp32 precision.
// gfx950 has no TF32-like mode; v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain at the
// vector-unit rate, so the kernel is written to be MFMA-issue-bound:
//   * 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave a 64x64
//     sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator registers);
//   * K is consumed 32 at a time through double-buffered LDS (2 x 32 KiB, two workgroups
//     per CU); global loads for step k+1 are issued before the MFMAs of step k and written
//     to the other LDS buffer after them, one barrier per step;
//   * an operand whose contraction index is contiguous in memory is kept as [row][32 k] in
//     LDS (128-byte rows, 16-byte chunks XOR-swizzled by (row>>1)&7) and fetched with one
//     ds_read_b128 per four MFMAs: the two k-slots of the instruction are fed k = 8g+j and
//     k = 8g+4+j, a permutation of the contraction order that is legal because both
//     operands use it;
//   * an operand whose contraction index is the slow dimension ([k][row] in memory) is kept
//     as it is, and consecutive lanes read consecutive words (conflict-free ds_read_b32);
//   * workgroup ids are remapped so that each XCD (private L2) owns a contiguous run of
//     output tiles that share A panels.
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kThreads = 256;
constexpr int kTileFloats = BM * BK;  // 4096 floats = 16 KiB per operand per stage

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == kActRelu) return v > 0.f ? v : 0.f;
  if (act == kActGelu) return gelu_fwd(v);
  return v;
}

// Float offset of (row, 16-byte chunk c) in the swizzled [128][32] image.
__device__ __forceinline__ int kmajor_off(int row, int c) { return row * BK + ((c ^ ((row >> 1) & 7)) << 2); }

// Loads this thread's four 16-byte pieces of one operand tile into registers.
//   KMAJOR : tile is rows [r0, r0+128) x k [k0, k0+32) of a [R][ld] matrix (k contiguous)
//   !KMAJOR: tile is k [k0, k0+32) x rows [r0, r0+128) of a [K][ld] matrix (rows contiguous)
// ALIGNED: base address, leading dimension and contiguous extent are multiples of 4 floats, so a 4-float piece is
// one aligned 16-byte load or wholly outside the matrix.  Otherwise the piece is gathered float by float with a
// bounds check each (the tail path for odd shapes).
template <bool KMAJOR, bool ALIGNED>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, long long ld, long long r0, long long R, long long k0,
                                          long long K, f32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + i * kThreads;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (KMAJOR) {
      const long long row = r0 + (id >> 3), k = k0 + ((id & 7) << 2);
      if (ALIGNED) {
        if (row < R && k < K) v = *reinterpret_cast<const f32x4*>(base + row * ld + k);
      } else if (row < R) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < K) v[j] = base[row * ld + k + j];
      }
    } else {
      const long long k = k0 + (id >> 5), row = r0 + ((id & 31) << 2);
      if (ALIGNED) {
        if (k < K && row < R) v = *reinterpret_cast<const f32x4*>(base + k * ld + row);
      } else if (k < K) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (row + j < R) v[j] = base[k * ld + row + j];
      }
    }
    regs[i] = v;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(float* lds, const f32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + i * kThreads;
    if (KMAJOR) *reinterpret_cast<f32x4*>(lds + kmajor_off(id >> 3, id & 7)) = regs[i];
    else *reinterpret_cast<f32x4*>(lds + (id << 2)) = regs[i];
  }
}

// Fragment of one 32-row MFMA tile for k-group g (8 k-values): element j feeds MFMA step j.
template <bool KMAJOR>
__device__ __forceinline__ f32x4 read_frag(const float* lds, int tile_row0, int g, int lane) {
  const int r = lane & 31, h = lane >> 5;
  if (KMAJOR) {
    return *reinterpret_cast<const f32x4*>(lds + kmajor_off(tile_row0 + r, 2 * g + h));
  } else {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lds[(8 * g + 4 * h + j) * BM + tile_row0 + r];
    return v;
  }
}

template <bool A_KMAJOR, bool B_KMAJOR, bool ALIGNED>
__global__ __launch_bounds__(kThreads, 2) void gemm_f32_kernel(GemmArgs p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // XCD-aware remap (bijective for any grid size): ids congruent mod 8 share an XCD.
  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wgid / tiles_n, tn = wgid % tiles_n;
  const long long m0 = static_cast<long long>(tm) * BM, n0 = static_cast<long long>(tn) * BN;

  const float* A = static_cast<const float*>(p.A);
  const float* B = static_cast<const float*>(p.B);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra[4], rb[4];
  const long long nk = (p.K + BK - 1) / BK;
  load_tile<A_KMAJOR, ALIGNED>(A, p.lda, m0, p.M, 0, p.K, ra);
  load_tile<B_KMAJOR, ALIGNED>(B, p.ldb, n0, p.N, 0, p.K, rb);
  store_tile<A_KMAJOR>(smem, ra);
  store_tile<B_KMAJOR>(smem + kTileFloats, rb);
  __syncthreads();

  int cur = 0;
  for (long long kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      load_tile<A_KMAJOR, ALIGNED>(A, p.lda, m0, p.M, (kt + 1) * BK, p.K, ra);
      load_tile<B_KMAJOR, ALIGNED>(B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, rb);
    }
    const float* as = smem + cur * 2 * kTileFloats;
    const float* bs = as + kTileFloats;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = read_frag<A_KMAJOR>(as, wm + 32 * i, g, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read_frag<B_KMAJOR>(bs, wn + 32 * j, g, lane);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }
    if (more) {
      float* an = smem + (cur ^ 1) * 2 * kTileFloats;
      store_tile<A_KMAJOR>(an, ra);
      store_tile<B_KMAJOR>(an + kTileFloats, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // Epilogue.  Accumulator register e of lane l holds C[row = (e&3) + 8*(e>>2) + 4*(l>>5)][col = l&31].
  float* C = static_cast<float*>(p.C);
  float* P = static_cast<float*>(p.pre_act);
  const float* bias = static_cast<const float*>(p.bias);
  const int col_in = lane & 31, half = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long col = n0 + wn + 32 * j + col_in;
    if (col >= p.N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row >= p.M) continue;
        float v = p.alpha * acc[i][j][e] + bv;
        if (p.beta != 0.f) v += p.beta * C[row * p.ldc + col];
        if (P) P[row * p.ldc + col] = v;
        C[row * p.ldc + col] = act_fn(v, p.act);
      }
    }
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace {
// 16-byte global loads: the contiguous dimension of each operand, its leading dimension and its base address
// are multiples of 4 floats.
bool vector_aligned(const GemmArgs& a) {
  if (a.lda % 4 || a.ldb % 4) return false;
  if (a.a_kmajor ? (a.K % 4) : (a.M % 4)) return false;
  if (a.b_kmajor ? (a.K % 4) : (a.N % 4)) return false;
  if (a.A && !aligned16(a.A)) return false;
  if (a.B && !aligned16(a.B)) return false;
  return true;
}
}  // namespace

bool gemm_f32_supported(const GemmArgs& a) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
  const long long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  return tiles < (1ll << 31);
}

void launch_gemm_f32(const GemmArgs& a, hipStream_t stream) {
  const int tiles_m = static_cast<int>((a.M + BM - 1) / BM), tiles_n = static_cast<int>((a.N + BN - 1) / BN);
  const dim3 grid(static_cast<unsigned>(tiles_m) * static_cast<unsigned>(tiles_n));
  const size_t lds = 4 * kTileFloats * sizeof(float);
  const bool vec = vector_aligned(a);
  auto go = [&](auto ak, auto bk) {
    constexpr bool AK = decltype(ak)::value, BKM = decltype(bk)::value;
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, true>), grid, dim3(kThreads), lds, stream, a, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, false>), grid, dim3(kThreads), lds, stream, a, tiles_m, tiles_n);
  };
  if (a.a_kmajor && a.b_kmajor) go(std::true_type{}, std::true_type{});
  else if (a.a_kmajor && !a.b_kmajor) go(std::true_type{}, std::false_type{});
  else if (!a.a_kmajor && a.b_kmajor) go(std::false_type{}, std::true_type{});
  else go(std::false_type{}, std::false_type{});
}

}  // namespace kernels
}  // namespace dpa
