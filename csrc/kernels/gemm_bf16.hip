// bf16 matrix products on the gfx950 matrix cores (fp32 accumulate), fused bias /
// activation epilogue, all three operand layouts a linear layer needs.
//
// Serves the same nn.Linear forward / backward products as gemm_f32.hip
// (reference min_DDP.py:41-49, 98, 103) for the bf16 configurations (wide MLP, GPT-2).
//   * 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 =
//     4x4 tiles of v_mfma_f32_16x16x32_bf16 (64 accumulator registers);
//   * K is consumed 64 at a time through double-buffered LDS (2 x 32 KiB); the global loads
//     of step k+1 are in flight during the MFMAs of step k and land in the other buffer
//     after them, one barrier per step;
//   * an operand with the contraction index contiguous in memory is kept as [row][64 k]
//     (128-byte rows, 16-byte chunks XOR-swizzled by (row>>1)&7): one ds_read_b128 is one
//     whole MFMA fragment (8 consecutive k of one row);
//   * an operand stored [k][row] (weights in the input-gradient product, both operands of
//     the weight-gradient product) is kept as [64 k][128 rows] and read with the hardware
//     transposing load ds_read_b64_tr_b16: a 16-lane group fetches a 4(k) x 16(row) block
//     and each lane receives its row's four k values, so no transpose pass over memory and
//     no scattered LDS writes are needed.  16-byte chunks are XOR-swizzled by
//     2*((k&3) + 4*((k>>3)&1)) so that the eight 4-row blocks a 32-lane half touches fall
//     on disjoint bank ranges;
//   * XCD-aware workgroup remap as in the fp32 kernel.
#include <atomic>
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kTileBytes = BM * BK * 2;  // 16 KiB per operand per stage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == kActRelu) return v > 0.f ? v : 0.f;
  if (act == kActGelu) return gelu_fwd(v);
  return v;
}

// Epilogue for one 16x16 accumulator tile computed with SWAPPED MFMA operands (weight fragment
// first): register e of lane l then holds C[row = m_base + (l&15)][col = n_base + 4*(l>>4) + e],
// i.e. four consecutive output columns, which leave the lane as ONE 8-byte store instead of four
// scattered 2-byte stores.
template <typename TR = BF16>
__device__ __forceinline__ void store_tile_bf16(const GemmArgs& p, const f32x4& acc, long long row, long long col0) {
  if (row >= p.M || col0 >= p.N) return;
  u16* C = static_cast<u16*>(p.C);
  u16* P = static_cast<u16*>(p.pre_act);
  const u16* bias = static_cast<const u16*>(p.bias);
  const bool full = (col0 + 4 <= p.N) && ((p.ldc & 3) == 0);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = p.alpha * acc[e];
  if (full) {
    if (bias) {
      const u16x4 b = *reinterpret_cast<const u16x4*>(bias + col0);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += TR::to_f32(b[e]);
    }
    u16* cp = C + row * p.ldc + col0;
    if (p.beta != 0.f) {
      const u16x4 old = *reinterpret_cast<const u16x4*>(cp);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += p.beta * TR::to_f32(old[e]);
    }
    if (P) {
      u16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = TR::from_f32(v[e]);
      *reinterpret_cast<u16x4*>(P + row * p.ldc + col0) = o;
    }
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = TR::from_f32(act_fn(TR::to_f32(TR::from_f32(v[e])), p.act));  // activation of the value as stored
    *reinterpret_cast<u16x4*>(cp) = o;
  } else {
    for (int e = 0; e < 4 && col0 + e < p.N; ++e) {
      float x = v[e] + (bias ? TR::to_f32(bias[col0 + e]) : 0.f);
      if (p.beta != 0.f) x += p.beta * TR::to_f32(C[row * p.ldc + col0 + e]);
      if (P) P[row * p.ldc + col0 + e] = TR::from_f32(x);
      C[row * p.ldc + col0 + e] = TR::from_f32(act_fn(TR::to_f32(TR::from_f32(x)), p.act));
    }
  }
}

// Byte offset of 16-byte chunk c (8 k-values) of `row` in the swizzled [128][64] image.
__device__ __forceinline__ int kmajor_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// Byte offset of 16-byte chunk c (8 rows) of contraction index k in the swizzled [64][128] image.
__device__ __forceinline__ int rmajor_off(int k, int c) { return k * 256 + ((c ^ (2 * ((k & 3) + 4 * ((k >> 3) & 1)))) << 4); }

// ALIGNED: every 8-element piece is either wholly inside the matrix and 16-byte aligned or wholly outside (base
// address, leading dimension and contiguous extent are multiples of 8 elements): one 16-byte load per piece.
// Otherwise (odd extents, leading dimensions or base addresses) the piece is gathered element by element with a
// bounds check each -- the tail path that lets every shape run here; it costs load instructions, not correctness.
template <bool KMAJOR, bool ALIGNED>
__device__ __forceinline__ void load_tile(const u16* __restrict__ base, long long ld, long long r0, long long R, long long k0,
                                          long long K, u32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + i * kThreads;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (KMAJOR) {
      const long long row = r0 + (id >> 3), k = k0 + ((id & 7) << 3);
      if (ALIGNED) {
        if (row < R && k < K) v = *reinterpret_cast<const u32x4*>(base + row * ld + k);
      } else if (row < R && k < K) {
        const u16* src = base + row * ld + k;
        const int n = K - k < 8 ? static_cast<int>(K - k) : 8;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = j < n ? src[j] : static_cast<unsigned short>(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = static_cast<unsigned>(e[2 * j]) | (static_cast<unsigned>(e[2 * j + 1]) << 16);
      }
    } else {
      const long long k = k0 + (id >> 4), row = r0 + ((id & 15) << 3);
      if (ALIGNED) {
        if (k < K && row < R) v = *reinterpret_cast<const u32x4*>(base + k * ld + row);
      } else if (k < K && row < R) {
        const u16* src = base + k * ld + row;
        const int n = R - row < 8 ? static_cast<int>(R - row) : 8;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = j < n ? src[j] : static_cast<unsigned short>(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = static_cast<unsigned>(e[2 * j]) | (static_cast<unsigned>(e[2 * j + 1]) << 16);
      }
    }
    regs[i] = v;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(char* lds, const u32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + i * kThreads;
    const int off = KMAJOR ? kmajor_off(id >> 3, id & 7) : rmajor_off(id >> 4, id & 15);
    *reinterpret_cast<u32x4*>(lds + off) = regs[i];
  }
}

// MFMA fragment of the 16-row tile starting at tile_row0 for k-substep kk (32 k-values):
// lane l holds row l&15, k = 32*kk + 8*(l>>4) + 0..7.
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int tile_row0, int kk, int lane) {
  if (KMAJOR) {
    return *reinterpret_cast<const bf16x8*>(lds + kmajor_off(tile_row0 + (lane & 15), 4 * kk + (lane >> 4)));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int k1 = 32 * kk + 8 * g + q;
    const int c = (tile_row0 >> 3) + (pp >> 1);
    const int sub = (pp & 1) << 3;
    using lds_ptr = __attribute__((address_space(3))) s16x4*;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + rmajor_off(k1, c) + sub));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + rmajor_off(k1 + 4, c) + sub));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

// TR = BF16 or F16: the two 16-bit types share everything but the MFMA opcode and the epilogue's conversions.
template <bool A_KMAJOR, bool B_KMAJOR, bool ALIGNED, typename TR>
__global__ __launch_bounds__(kThreads, 2) void gemm_bf16_kernel(GemmArgs p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem_bf16[];

  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = wgid / tiles_n, tn = wgid % tiles_n;
  const long long m0 = static_cast<long long>(tm) * BM, n0 = static_cast<long long>(tn) * BN;

  const u16* A = static_cast<const u16*>(p.A);
  const u16* B = static_cast<const u16*>(p.B);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  const long long nk = (p.K + BK - 1) / BK;
  load_tile<A_KMAJOR, ALIGNED>(A, p.lda, m0, p.M, 0, p.K, ra);
  load_tile<B_KMAJOR, ALIGNED>(B, p.ldb, n0, p.N, 0, p.K, rb);
  store_tile<A_KMAJOR>(smem_bf16, ra);
  store_tile<B_KMAJOR>(smem_bf16 + kTileBytes, rb);
  __syncthreads();

  int cur = 0;
  for (long long kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      load_tile<A_KMAJOR, ALIGNED>(A, p.lda, m0, p.M, (kt + 1) * BK, p.K, ra);
      load_tile<B_KMAJOR, ALIGNED>(B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, rb);
    }
    const char* as = smem_bf16 + cur * 2 * kTileBytes;
    const char* bs = as + kTileBytes;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<A_KMAJOR>(as, wm + 16 * i, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<B_KMAJOR>(bs, wn + 16 * j, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (std::is_same<TR, F16>::value) {
            typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa[i]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
          }
        }
    }
    if (more) {
      char* an = smem_bf16 + (cur ^ 1) * 2 * kTileBytes;
      store_tile<A_KMAJOR>(an, ra);
      store_tile<B_KMAJOR>(an + kTileBytes, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // Epilogue (operands were swapped in the MFMA: see store_tile_bf16).
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      store_tile_bf16<TR>(p, acc[i][j], m0 + wm + 16 * i + (lane & 15), n0 + wn + 16 * j + 4 * (lane >> 4));
}

// =====================================================================================
// Large-tile kernel: 256x256 output tile, 512 threads (8 waves as 2(M) x 4(N), each wave
// 128x64 = 8x4 MFMA tiles, 128 accumulator registers), K consumed 64 at a time.
//   * operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR staging and
//     no ds_write instructions, which is what bounds the 128x128 kernel above (ds_write_b128
//     moves ~79 B/clk/CU, less than a third of the read rate);
//   * the DMA writes LDS linearly (wave base + lane*16), so the bank-conflict swizzle is applied
//     to each lane's GLOBAL source address, and the same XOR is applied when fragments are read;
//   * a k-step of an operand is staged as two HALF-TILES of 128 rows x 64 k (16 KiB, two DMA
//     instructions per wave).  Half h of A holds rows wr*128 + h*64 .. +63 of both wave rows wr,
//     half h of B holds rows wc*64 + h*32 .. +31 of all four wave columns wc, so every wave reads
//     each half-tile in exactly one of the four phases of a k-step and the half-tile can be
//     restaged two phases later;
//   * two k-steps of LDS (2 x 64 KiB, one workgroup per CU).  Half-tiles are requested in the order
//     B0 A0 B1 A1, one per phase, three half-tiles (six DMA instructions per wave) stay in flight
//     across the barriers: the only wait inside the loop is one counted s_waitcnt vmcnt(6) per k-step;
//   * a k-step is four phases of 16 MFMAs (one quadrant of the wave's tile: 64 rows x 32 columns x
//     64 k); each phase is {fragment reads + one half-tile request | barrier | MFMAs | barrier}, and
//     the upper wave row runs one barrier behind the lower one, so on every SIMD one wave issues
//     MFMAs while its partner reads LDS and requests memory;
//   * rows / columns past the matrix edge are clamped to the last valid one (their results are
//     never stored); K must be a multiple of 64 (no zero fill exists on the DMA path).
// =====================================================================================
// Diagnostic builds only (tools/kernel_harness, -DDPA_GEMM_STAMPS): wave 0 of every workgroup records
// the shader clock at a few points into p.colsum_partial (unused by the plain product).
#ifdef DPA_GEMM_STAMPS
#define DPA_STAMP(i)                                                                                         \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
    if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(p.colsum_partial)[blockIdx.x * 8 + (i)] = t_; \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#define DPA_STAMP_REAL(i)                                                                                    \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(p.colsum_partial)[blockIdx.x * 8 + (i)] = t_; \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#else
#define DPA_STAMP(i)
#define DPA_STAMP_REAL(i)
#endif
// -DGEMM_TILE_TIMES (tools/kernel_harness/gemm_bench.hip): wave 0 records the 100 MHz clock at the start (slot 0) and
// behind the last store (slot 1) of each of its first 128 tiles into p.colsum_partial (eight slots per tile: 2 = k-steps
// done, 3 .. 6 = stations of the activation-backward epilogue, 7 = kernel entry).
#ifdef GEMM_TILE_TIMES
#define DPA_TILE_TIME(slot)                                                                                          \
  do {                                                                                                               \
    unsigned long long t_;                                                                                           \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                 \
    if (threadIdx.x == 0 && tile_no < 128)                                                                           \
      reinterpret_cast<unsigned long long*>(p.colsum_partial)[(blockIdx.x * 128ll + tile_no) * 8 + (slot)] = t_;     \
  } while (0)
#else
#define DPA_TILE_TIME(slot)
#endif

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int kThreads2 = 512;
constexpr int kTileBytes2 = BM2 * BK2 * 2;  // 32 KiB per operand per k-step
constexpr int kHalfBytes = 128 * BK2 * 2;   // 16 KiB: one half-tile
// position of the half-tiles inside a k-step's 64 KiB, in request order
constexpr int kHalfB0 = 0, kHalfA0 = 1, kHalfB1 = 2, kHalfA1 = 3;

// [128 local rows][64 k] image, 128-byte rows, 16-byte chunks XOR-swizzled by (row>>1)&7: the 16
// lanes a ds_read_b128 serves per LDS cycle (rows 0-3 and 12-15 at one chunk, rows 4-11 at the next)
// then fall on 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int kmajor2_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// [64 k][128 local rows] image, 256-byte rows (one bank row each), 16 chunks: the eight k-rows a
// 32-lane half of a transposing read touches (q and 8+q, q = 0..3) get eight different 32-byte slots.
__device__ __forceinline__ int rmajor2_off(int k, int c) { return k * 256 + ((c ^ (2 * ((k & 3) + 4 * ((k >> 3) & 1)))) << 4); }

// GELU'(x) of a bf16 x only takes 65536 values, and outside 2^-12 <= |x| < 8 it is 0.5, 0 or 1 to well
// below bf16 resolution (likewise Phi(x) for the forward, GELU(x) = x * Phi(x)): the epilogues look it up in an LDS table
// (15 binades x 128 mantissas per sign, fp32 entries computed by the workgroup itself while its first
// tile is in flight) instead of evaluating exp / rcp / eight FMAs per element -- with one workgroup per
// CU nothing overlaps the epilogue's VALU work, which used to take as long as nine k-steps.
constexpr int kGeluLo = 0x3980;     // bf16 bits of 2^-12
constexpr int kGeluEntries = 1920;  // up to 0x40FF = 7.97
constexpr int kGeluTableBytes = 2 * kGeluEntries * 4;

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

// This wave's share of one half-tile is 2 DMA instructions of 1 KiB each (pieces 2*wave, 2*wave+1 of
// 16).  Their per-lane source addresses are split into a wave-uniform 64-bit base that moves with
// the k-step (kept in SGPRs, advanced by scalar adds) and a per-lane 32-bit byte offset inside the
// tile's rows that never changes, computed once here: the loop body then spends no vector registers
// or VALU work on addresses.  Offsets stay far below 4 GiB because they are relative to the tile
// (256 rows at most for a k-contiguous operand, 64 k-rows otherwise), not to the matrix.
// SPAN = rows one wave row / column owns inside a half-tile (64 for A, 32 for B); the same wave
// row / column owns 2*SPAN consecutive rows of the full tile.
template <bool KMAJOR, int SPAN>
__device__ __forceinline__ void dma_offsets(long long ld, long long r0, long long R, int wid, int lane, unsigned (&off)[2][2]) {
  // 32-bit arithmetic throughout (the cursor moves to a new tile in the middle of a k-step, next to 180
  // live fragment and accumulator registers): a tile-relative row < 256 times a row pitch < 8 MiB.
  const unsigned ld2 = static_cast<unsigned>(ld) * 2u;
  // last valid row (k-contiguous operand) or last valid 8-row chunk start (the dispatcher requires R % 8 == 0), tile-relative
  const long long room = R - r0 - (KMAJOR ? 1 : 8);
  const unsigned lim = static_cast<unsigned>(room < 255 ? room : 255);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wid * 2 + i;  // 16 pieces of 1 KiB per half-tile
      if (KMAJOR) {
        const int lr = piece * 8 + (lane >> 3);                   // 8 local rows of 128 B per piece
        const int c = (lane & 7) ^ ((lr >> 1) & 7);               // logical chunk that belongs at this physical slot
        unsigned rel = static_cast<unsigned>((lr / SPAN) * (2 * SPAN) + h * SPAN + lr % SPAN);
        rel = rel < lim ? rel : lim;                              // rows past the edge re-read the last row
        off[h][i] = rel * ld2 + static_cast<unsigned>(c * 16);
      } else {
        const int k = piece * 4 + (lane >> 4);                    // 4 k-rows of 256 B per piece
        const int c = (lane & 15) ^ (2 * ((k & 3) + 4 * ((k >> 3) & 1)));
        const int lr = c * 8;                                     // first of the chunk's 8 local rows
        unsigned rel = static_cast<unsigned>((lr / SPAN) * (2 * SPAN) + h * SPAN + lr % SPAN);
        rel = rel < lim ? rel : lim;
        off[h][i] = static_cast<unsigned>(k) * ld2 + rel * 2u;
      }
    }
}
// Base of the tile for k-step starting at k0 (wave-uniform).
template <bool KMAJOR>
__device__ __forceinline__ const char* dma_base(const u16* base, long long ld, long long r0, long long k0) {
  return reinterpret_cast<const char*>(KMAJOR ? base + r0 * ld + k0 : base + k0 * ld + r0);
}
// Implicit 3x3 convolution, operand A (conv_mode 1): tile-relative row -> output pixel (b, h, w) -> byte offset of
// the haloed pixel (b, h, w) -- the top-left tap -- from X's origin, plus the lane's swizzled 16-byte chunk.  The
// k-steps then walk (dh, dw, c) by moving the wave-uniform base only: 3*C consecutive elements per kernel row.
__device__ __forceinline__ void dma_offsets_conv_a(const GemmArgs& p, long long r0, int wid, int lane, unsigned (&off)[2][2]) {
  const long long room = p.M - r0 - 1;
  const unsigned lim = static_cast<unsigned>(room < 255 ? room : 255);
  const unsigned H = static_cast<unsigned>(p.conv_H), W = static_cast<unsigned>(p.conv_W), HW = H * W;
  const unsigned pix_bytes = static_cast<unsigned>(p.conv_C) * 2u;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wid * 2 + i;
      const int lr = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((lr >> 1) & 7);
      unsigned rel = static_cast<unsigned>((lr / 64) * 128 + h * 64 + lr % 64);
      rel = rel < lim ? rel : lim;
      const unsigned m = static_cast<unsigned>(r0) + rel;  // M < 2^31 (checked by the host)
      const unsigned b = m / HW, rem = m - b * HW, hh = rem / W, ww = rem - hh * W;
      const unsigned pix = (b * (H + 2u) + hh) * (W + 2u) + ww;
      off[h][i] = pix * pix_bytes + static_cast<unsigned>(c * 16);
    }
}
// Operand B of the weight gradient (conv_mode 2): column n = (3*dh + dw)*C + c of the [K][N] operand lives at
// X row (q + dh*(W+2) + dw - (W+3)), channel c; the "- (W+3)" is folded into the wave-uniform base.
__device__ __forceinline__ void dma_offsets_conv_b(const GemmArgs& p, long long r0, int wid, int lane, unsigned (&off)[2][2]) {
  const long long room = p.N - r0 - 8;
  const unsigned lim = static_cast<unsigned>(room < 255 ? room : 255);
  const unsigned C = static_cast<unsigned>(p.conv_C), W2 = static_cast<unsigned>(p.conv_W) + 2u;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wid * 2 + i;
      const int k = piece * 4 + (lane >> 4);
      const int c = (lane & 15) ^ (2 * ((k & 3) + 4 * ((k >> 3) & 1)));
      const int lr = c * 8;
      unsigned rel = static_cast<unsigned>((lr / 32) * 64 + h * 32 + lr % 32);
      rel = rel < lim ? rel : lim;
      const unsigned n = static_cast<unsigned>(r0) + rel;
      const unsigned dh = n / (3u * C), within = n - dh * 3u * C;   // (dw, c) are contiguous in X
      off[h][i] = (static_cast<unsigned>(k) * C + dh * W2 * C + within) * 2u;
    }
}
// Issued as inline assembly: the compiler orders every LDS read it cannot prove disjoint from an
// LDS-DMA it knows about (the transposing reads always) behind s_waitcnt vmcnt(0), which would put
// the whole DMA of tile k+1 in front of the math of tile k.  The kernel orders DMA and reads itself
// (explicit vmcnt wait + barrier).  The hardware adds lane * 16 to the LDS address in M0.
// The tile base is wave-uniform by construction; the explicit scalarisation keeps it in SGPRs even when
// the compiler evaluates the 64-bit address arithmetic on the vector ALU (the "s" constraint of the
// assembly below does not convert by itself).
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ void dma_issue(const char* base, const unsigned (&off)[2], unsigned lds_dst) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned dst = lds_dst + i * 1024;
    // s_nop 4: the base may have reached its SGPRs through a VALU instruction (v_readfirstlane, v_readlane of a
    // spilled scalar) right in front of this statement, and a vector-memory instruction must not read such
    // an SGPR for five wait states; the compiler pads that for its own instructions only.
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off[i]), "s"(base), "s"(dst) : "memory");
  }
}

// Fragment reads of the large-tile kernel work on LDS byte addresses held in registers: one set per
// operand, computed once, valid for LDS buffer 0 and flipped to the other buffer (bit 16) after every
// k-step, so that a phase's reads are nothing but ds_read instructions with immediate offsets.
//   k-contiguous image: addr[kk] (kk = 0, 1) is this lane's 16 bytes of 16-row tile 0; tile t lies
//     t * 2048 bytes further (the swizzle only involves the row inside the tile).
//   transposed image: addr[t] is this lane's first 8 bytes of tile t for kk = 0 (the swizzle XORs the
//     tile index into the chunk, so tiles are not a constant apart); the second k-block of four lies
//     1024 bytes, kk = 1 lies 8192 bytes further.
template <bool KMAJOR>
struct FragAddr {
  unsigned addr[KMAJOR ? 2 : 4];
};
template <bool KMAJOR>
__device__ __forceinline__ void frag_addr_init(FragAddr<KMAJOR>& fa, unsigned lds_base, int wave_row0, int lane) {
  if constexpr (KMAJOR) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa.addr[kk] = lds_base + kmajor2_off(wave_row0 + (lane & 15), 4 * kk + (lane >> 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
    for (int t = 0; t < 4; ++t) fa.addr[t] = lds_base + rmajor2_off(8 * g + q, ((wave_row0 + 16 * t) >> 3) + (pp >> 1)) + ((pp & 1) << 3);
  }
}
template <bool KMAJOR>
__device__ __forceinline__ void frag_addr_flip(FragAddr<KMAJOR>& fa) {
#pragma unroll
  for (int t = 0; t < (KMAJOR ? 2 : 4); ++t) {
    fa.addr[t] ^= 2 * kTileBytes2;
    asm volatile("" : "+v"(fa.addr[t]));  // keep the flipped address in its register instead of re-deriving it in front of the reads
  }
}
// Fragment of 16-row tile t, k-substep kk, out of the half-tile image at byte offset HALF_OFF of the buffer.
template <bool KMAJOR, int HALF_OFF>
__device__ __forceinline__ bf16x8 read_frag2(const FragAddr<KMAJOR>& fa, int t, int kk) {
  if constexpr (KMAJOR) {
    using lds_ptr = const __attribute__((address_space(3))) bf16x8*;
    return *(lds_ptr)(fa.addr[kk] + HALF_OFF + t * 2048);
  } else {
    using lds_ptr = __attribute__((address_space(3))) s16x4*;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(fa.addr[t] + HALF_OFF + kk * 8192));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(fa.addr[t] + HALF_OFF + kk * 8192 + 1024));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

// LDS map of the large-tile kernel: two k-steps of operands, the GELU table, a 2 KiB epilogue staging
// strip per wave, one control word (the id of the next dynamically assigned tile) and a tile's bias values.
constexpr int kRingBytes = 4 * kTileBytes2;                 // 128 KiB
constexpr int kStageOff = kRingBytes + kGeluTableBytes;     // 16 rows x 128 B per wave
constexpr int kCtrlOff = kStageOff + 8 * 2048;
constexpr int kBiasOff = kCtrlOff + 64;                     // the next tile's 256 bias values (bf16)
constexpr int kLdsBytes2 = kBiasOff + 512;                  // 163,392 of the CU's 163,840 bytes
constexpr int kNoTile = 0x7fffffff;
using lds_u32_ptr = volatile __attribute__((address_space(3))) unsigned*;
// The lane index, recomputed (two instructions) where the per-tile code needs it.  Kept in a register from the top of
// the kernel it does not survive the tile loop's register pressure: it is spilled, and a reload from scratch memory
// is an ordinary memory operation -- the compiler waits for it with vmcnt(0), which also waits for every LDS-DMA
// request and output store in flight (found twice per tile, in the cursor move and in front of the bias request).
// The two GELU tables (index 0: Phi(x) for the forward epilogue, 1: GELU'(x) for the backward one), filled once per
// device by gelu_tables_init_kernel; a workgroup copies the one it needs into LDS by LDS-DMA.  (Computing the 3840
// entries in every workgroup took 3.6 us -- a tenth of a tile's time in the one-tile-per-workgroup backward epilogue.)
__device__ float g_gelu_tables[2][2 * kGeluEntries];
__global__ void gelu_tables_init_kernel() {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * kGeluEntries) return;
  const bool neg = e >= kGeluEntries;
  const unsigned bits = static_cast<unsigned>(kGeluLo + (neg ? e - kGeluEntries : e)) | (neg ? 0x8000u : 0u);
  const float x = __uint_as_float(bits << 16);
  g_gelu_tables[0][e] = normal_cdf(x);
  g_gelu_tables[1][e] = gelu_grad(x);
}

__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
constexpr int kStreamingMaxKSteps = 64;  // beyond K = 4096 a tile is long enough to hide its own start

// SPLIT: the grid is (tiles, ranges); every workgroup multiplies one contraction range of `k_iters` steps of one
// tile; raw fp32 partial tiles go to p.workspace[range][M][N] (16-byte stores) and splitk_reduce_kernel finishes
// the product.
//
// tile_counter == nullptr: one tile per workgroup (grid = number of tiles, XCD-aware order).
// tile_counter != nullptr: the grid is at most one workgroup per CU and every workgroup walks a
// sequence of tiles -- blockIdx.x, blockIdx.x + gridDim.x, then ids drawn from the counter (which the
// launcher zeroes) -- as ONE stream of k-steps: the half-tile requests run on into the next tile while
// the current one is finished, so that only the first tile of a workgroup pays the memory latency of a
// cold start and the next tile's operands are in LDS when the epilogue ends.  Drawing ids instead of
// fixing them per workgroup keeps the kernel balanced when another stream (a collective) holds some CUs.
// CONV: 0 = plain product; 1 = operand A is the implicit 3x3-convolution view of a haloed image (NT layout);
// 2 = operand B is that view for the weight gradient (split-K TN layout).  See GemmArgs::conv_mode.
// ACTBWD: the activation-backward epilogue (GemmArgs::aux); an instantiation of its own, so that the plain products do
// not carry its code and registers (as a run-time branch it cost the NT forward kernel 8 spilled registers).
template <bool A_KMAJOR, bool B_KMAJOR, bool SPLIT, int CONV = 0, bool ACTBWD = false>
__global__ __launch_bounds__(kThreads2, 2) void gemm_bf16_8ph_kernel(GemmArgs p, int tiles_m, int tiles_n, int k_iters, unsigned* tile_counter) {
  static_assert(CONV == 0 || (CONV == 1 && A_KMAJOR && B_KMAJOR && !SPLIT) || (CONV == 2 && !A_KMAJOR && !B_KMAJOR && SPLIT),
                "convolution views exist for the NT forward product and the split-K TN weight gradient");
  static_assert(!ACTBWD || (!SPLIT && CONV == 0), "the activation-backward epilogue belongs to the plain single-pass product");
  extern __shared__ __attribute__((aligned(16))) char smem2[];  // [2 k-steps][B0 | A0 | B1 | A1][GELU table][staging][control]
  DPA_STAMP(0);
  DPA_STAMP_REAL(4);
  [[maybe_unused]] int tile_no = 0;
  DPA_TILE_TIME(7);

  float* gelu_tab = reinterpret_cast<float*>(smem2 + kRingBytes);

  const bool dynamic = !SPLIT && tile_counter != nullptr;
  const int ntiles = tiles_m * tiles_n;
  // SPLIT: the (range, tile) pairs are dealt to the XCDs as one list, range-major: workgroup L of the launch
  // (XCD L % 8) takes a contiguous eighth of it, so an XCD's ~32 workgroups multiply the SAME contraction range of
  // neighbouring tiles and share every operand panel through their L2.  Taken per range instead (tile from
  // blockIdx.x, range from blockIdx.y) an XCD holds 4-5 tiles of each of the ~7 ranges and fetches ~2.3x the panels.
  int split_idx = 0, first_id = blockIdx.x;
  if (SPLIT) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, W = gridDim.x * gridDim.y;
    const int q = W >> 3, r = W & 7, x = L & 7;
    const int w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (L >> 3);
    split_idx = w / ntiles;
    first_id = w - split_idx * ntiles;
  }
  // Tile id -> origin.  Ids are laid out in groups of 8 row-tiles x all column-tiles, row-tile fastest: the
  // tiles being worked on at any time then form blocks that share operand panels in L2 (an 8 x 4 block
  // touches 12 panels per k-step instead of the 33 of a single column of tiles, +9 % on 8192^3).  With one
  // tile per workgroup each XCD (workgroups b, b+8, ...) additionally gets a contiguous run of ids.
  auto tile_origin = [&](int id, long long& m0, long long& n0) __attribute__((always_inline)) {
    if (!dynamic && !SPLIT) {
      const int q = ntiles >> 3, r = ntiles & 7, xcd = id & 7;
      id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int gsz = 8 * tiles_n;
    const int gid = id / gsz, first_m = gid * 8;
    const int gm = tiles_m - first_m < 8 ? tiles_m - first_m : 8;
    m0 = static_cast<long long>(first_m + (id % gsz) % gm) * BM2;
    n0 = static_cast<long long>((id % gsz) / gm) * BN2;
  };

  const u16* A = static_cast<const u16*>(p.A);
  const u16* B = static_cast<const u16*>(p.B);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int wm = wr * 128, wn = wc * 64;
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lptr_t)smem2));
  if (!SPLIT && (ACTBWD ? p.aux_mode == kActGelu : p.act == kActGelu)) {
    // backward epilogue: GELU'(x); forward epilogue: Phi(x), so that GELU(x) = x * table[x].  15 pieces of 1 KiB, two
    // per wave: older than every operand request, so the prologue's counted wait and barrier publish them as well.
    const char* src = reinterpret_cast<const char*>(g_gelu_tables[ACTBWD ? 1 : 0]);
    const unsigned voff = static_cast<unsigned>(lane * 16);
    for (int piece = wid; piece < kGeluTableBytes / 1024; piece += 8) {
      const char* base = uniform_ptr(src + piece * 1024);
      const unsigned dst = lds_base + kRingBytes + piece * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory");
    }
  }

  const long long nk_all = p.K / BK2;
  const long long kt0 = SPLIT ? static_cast<long long>(split_idx) * k_iters : 0;
  const long long nk_end = SPLIT ? (kt0 + k_iters < nk_all ? kt0 + k_iters : nk_all) : nk_all;
  const int n = static_cast<int>(nk_end - kt0);  // k-steps per tile

  // ---- request cursor: where the stream of half-tile requests stands (tile, k-step, LDS buffer).
  // It enters the next tile at a fixed point of the program -- between phases 1 and 2 of a tile's
  // second-to-last k-step -- so the steady-state k-steps carry no bookkeeping at all.
  const long long a_step = A_KMAJOR ? BK2 * 2 : BK2 * 2 * p.lda, b_step = B_KMAJOR ? BK2 * 2 : BK2 * 2 * p.ldb;  // bytes per k-step
  unsigned off_a[2][2], off_b[2][2];
  const char *r_aptr = nullptr, *r_bptr = nullptr;
  unsigned r_dst = lds_base + wid * 2048;  // this wave's first piece of half-tile 0 in the buffer being filled
  bool r_valid = true;
  // Dynamic walk: the id list is cut into eight contiguous segments, one per XCD (workgroup b runs on XCD b % 8).
  // The W = gridDim.x / 8 workgroups of an XCD take local indices j and W + j of their segment, then indices drawn
  // from the segment's own counter: at any time they multiply ~32 consecutive ids -- an 8 x 4 block of tiles, 12
  // operand panels per k-step through that XCD's L2.  (One counter for the whole chip scatters every such block
  // over all eight L2s: each then sees ~33 panels per k-step and misses 40 % of its requests.)  A workgroup whose
  // segment is used up moves on to the next XCD's, so the kernel stays balanced when some CUs are held elsewhere.
  const int seg_w = static_cast<int>(gridDim.x >> 3);
  int seg = static_cast<int>(blockIdx.x & 7), seg_hops = 0, seg_start = 0, seg_len = 0;
  auto seg_range = [&](int x) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7;
    seg_start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    seg_len = q + (x < r ? 1 : 0);
  };
  if (dynamic) {
    seg_range(seg);
    first_id = seg_start + static_cast<int>(blockIdx.x >> 3);  // the launcher keeps 2 W <= seg_len
  }
  int cur = first_id;                                                                              // tile being multiplied
  const int seg_static = 2 * seg_w;  // local indices handed out without a draw, per segment
  int nxt = dynamic ? seg_start + seg_w + static_cast<int>(blockIdx.x >> 3) : kNoTile;  // tile after it
  // per-lane source offsets of the cursor's tile (recomputed after every epilogue rather than kept across it)
  auto set_offsets = [&](int id) __attribute__((always_inline)) {
    long long rm0, rn0;
    tile_origin(id, rm0, rn0);
    // an opaque copy of the lane index: the compiler would otherwise hoist the lane-only parts of all eight
    // offsets to the top of the kernel and keep (spill) them across everything
    const int l = fresh_lane();
    if constexpr (CONV == 1) dma_offsets_conv_a(p, rm0, wid, l, off_a);
    else dma_offsets<A_KMAJOR, 64>(p.lda, rm0, p.M, wid, l, off_a);
    if constexpr (CONV == 2) dma_offsets_conv_b(p, rn0, wid, l, off_b);
    else dma_offsets<B_KMAJOR, 32>(p.ldb, rn0, p.N, wid, l, off_b);
  };
  // CONV == 1: the A base does not depend on the tile; it walks 128 bytes per k-step through the 3*C contiguous
  // elements of a kernel row and jumps to the next image row every 3*C/64 k-steps.
  const int conv_row_steps = CONV == 1 ? 3 * p.conv_C / BK2 : 0;
  const long long conv_row_jump = CONV == 1 ? static_cast<long long>(p.conv_W - 1) * p.conv_C * 2 : 0;
  int conv_step = 0;
  auto set_cursor = [&](int id) __attribute__((always_inline)) {
    long long rm0, rn0;
    tile_origin(id, rm0, rn0);
    set_offsets(id);
    if constexpr (CONV == 1) {
      r_aptr = uniform_ptr(reinterpret_cast<const char*>(A));
      conv_step = 0;
    } else {
      r_aptr = uniform_ptr(dma_base<A_KMAJOR>(A, p.lda, rm0, kt0 * BK2));
    }
    if constexpr (CONV == 2) {
      // k-step kt0 of X, shifted up by the (W + 3) rows that dma_offsets_conv_b adds back per column
      r_bptr = uniform_ptr(reinterpret_cast<const char*>(B + (kt0 * BK2 - (p.conv_W + 3)) * p.conv_C));
    } else {
      r_bptr = uniform_ptr(dma_base<B_KMAJOR>(B, p.ldb, rn0, kt0 * BK2));
    }
  };
  // Requests the next half-tile of the stream; `which` runs B0 A0 B1 A1 B0 ... by construction of the callers.
  auto request = [&](auto which_c) __attribute__((always_inline)) {
    constexpr int which = decltype(which_c)::value;
    if (which & 1) dma_issue(r_aptr, off_a[which >> 1], r_dst + which * kHalfBytes);
    else dma_issue(r_bptr, off_b[which >> 1], r_dst + which * kHalfBytes);
    if (which == kHalfA1) {  // k-step complete
      r_dst ^= 2 * kTileBytes2;
      if constexpr (CONV == 1) {
        r_aptr += BK2 * 2;
        if (++conv_step == conv_row_steps) {
          conv_step = 0;
          r_aptr += conv_row_jump;
        }
      } else {
        r_aptr += a_step;
      }
      r_bptr += b_step;
    }
  };
  constexpr std::integral_constant<int, kHalfB0> B0{};
  constexpr std::integral_constant<int, kHalfA0> A0{};
  constexpr std::integral_constant<int, kHalfB1> B1{};
  constexpr std::integral_constant<int, kHalfA1> A1{};
  // nothing may be scheduled across the barrier
  auto phase_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x4 acc[8][4];
  // Fragment registers: the A fragments of one 64-row half (reloaded in phase 3), the B fragments of
  // both 32-column halves (kept for the whole k-step: B0 is used again in phase 4).
  bf16x8 fa[2][4], fb0[2][2], fb1[2][2];
  int c_buf = 0;  // LDS buffer holding the k-step being multiplied
  FragAddr<A_KMAJOR> addr_a;
  FragAddr<B_KMAJOR> addr_b;
  frag_addr_init<A_KMAJOR>(addr_a, lds_base, wr * 64, lane);
  frag_addr_init<B_KMAJOR>(addr_b, lds_base, wc * 32, lane);
  auto read_a = [&](auto h_c) __attribute__((always_inline)) {
    constexpr int off = (decltype(h_c)::value ? kHalfA1 : kHalfA0) * kHalfBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[kk][i] = read_frag2<A_KMAJOR, off>(addr_a, i, kk);
  };
  auto read_b = [&](auto h_c, bf16x8 (&fb)[2][2]) __attribute__((always_inline)) {
    constexpr int off = (decltype(h_c)::value ? kHalfB1 : kHalfB0) * kHalfBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[kk][j] = read_frag2<B_KMAJOR, off>(addr_b, j, kk);
  };
  // 16 MFMAs: quadrant (row half ah, column half bh) of the wave's tile over the whole k-step
  auto multiply = [&](int ah, int bh, const bf16x8 (&fb)[2][2]) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * ah + i][2 * bh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[4 * ah + i][2 * bh + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // Activation-backward epilogue only (one tile per workgroup): the aux tile is fetched by LDS-DMA as well,
  // each wave's 128 x 64 piece in two halves of 64 rows x 128 B (8 KiB, 8 instructions) at
  // buffer * 64 KiB + wave * 8 KiB.  The first half goes into the buffer that the last k-step leaves idle,
  // so its latency hides behind that step's MFMAs; the second half follows once the operand buffers are
  // dead.  LDS position (lane & 7) of row (lane >> 3) receives chunk (lane & 7) ^ (lane >> 3) of the row
  // (XOR on the source address), which spreads the accumulator-layout reads of the epilogue over the banks.
  constexpr bool act_bwd = ACTBWD;
  long long m0, n0;  // origin of the tile being multiplied
  auto aux_request = [&](int half, int buf) __attribute__((always_inline)) {
    const int al = fresh_lane();
    const unsigned aux_off = static_cast<unsigned>(((al >> 3) * p.ldc + (((al & 7) ^ (al >> 3)) << 3)) * 2);
    const char* src = uniform_ptr(reinterpret_cast<const char*>(static_cast<const u16*>(p.aux) + (m0 + wm + 64 * half) * p.ldc + n0 + wn));
    const unsigned dst0 = lds_base + buf * 2 * kTileBytes2 + wid * 8192;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const char* base = uniform_ptr(src + it * 16 * p.ldc);  // 8 rows further
      const unsigned dst = dst0 + it * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(aux_off), "s"(base), "s"(dst) : "memory");
    }
  };

  unsigned fetched = 0;  // lane 0 of wave 0: id drawn from the tile counter (lands behind the k-step's vmcnt wait)
  // The bias is added by starting a tile's accumulators at the bias values of their columns instead of at
  // zero, so the epilogue is a plain conversion.  A tile's 256 values are fetched one tile ahead -- in front
  // of the prologue for a workgroup's first tile, in phase 1 of the previous tile's last k-step otherwise --
  // by wave 0 with two LDS-DMA instructions; they are older than the half-tiles the next counted wait leaves
  // in flight, so that wait (and the barrier behind it) publishes them like an operand tile.
  auto bias_request = [&](long long bn0) __attribute__((always_inline)) {
    if (wid != 0) return;
    const char* base = uniform_ptr(reinterpret_cast<const char*>(static_cast<const u16*>(p.bias) + bn0));
    const unsigned voff = static_cast<unsigned>(fresh_lane() * 4);
    const unsigned dst = lds_base + kBiasOff;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory");
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(base + 256), "s"(dst + 256) : "memory");
  };

  // One k-step = four phases; each phase is {fragment reads + one half-tile request | barrier | 16 MFMAs |
  // barrier}.  The request order and the reads' phases make every restage safe without further waits
  // ("retired" = the reads' data is in registers on every wave, which the barriers between guarantee):
  //   phase 1  reads B0 (retired before the phase's first barrier: lgkmcnt(8)) and A0   requests A1 of the next k-step -> other buffer
  //   phase 2  reads B1                                                                 requests B0 of the k-step after  -> this buffer
  //   phase 3  reads A1                                                                 requests A0 of the k-step after
  //   phase 4  (B0 still in registers)                                                  requests B1 of the k-step after, waits vmcnt(6)
  // A half-tile is requested again at the earliest two phases after its last read (B0: one phase, after
  // the early wait; the reads of a phase are complete when its MFMAs are, i.e. before its second barrier),
  // and the vmcnt(6) of phase 4 -- everything but the three newest half-tiles has landed, i.e. all of the
  // next k-step -- is followed by two barriers before phase 1 reads it.
  // FIRST:  a tile's first k-step.  When the tile follows another one of this workgroup, A1 of the second
  //         k-step was already requested in front of that tile's epilogue (a1_ahead), and the stores of
  //         that epilogue (ahead_stores of them) are younger than it: the wait of phase 4 then leaves
  //         them in flight as well instead of waiting for the output to reach memory.
  // PENULT: a tile's second-to-last k-step: draws the id of the tile after next, and moves the cursor
  //         into the next tile after the request of phase 1.
  // LAST:   a tile's last k-step: the cursor is in the next tile, if there is one; requests what the
  //         epilogue will need (bias, first half of the aux tile).
  // Neither: steady state (the cursor is inside this tile: every request is issued, nothing else happens).
  bool a1_ahead = false;
  int ahead_stores = 0;
  auto k_step = [&](auto first_c, auto penult_c, auto last_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value, PENULT = decltype(penult_c)::value, LAST = decltype(last_c)::value;
    // phase 1
    read_b(std::integral_constant<int, 0>{}, fb0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    if (PENULT && dynamic && threadIdx.x == 0) {
      const unsigned one = 1u, zero = 0u;
      asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(fetched) : "v"(zero), "v"(one), "s"(tile_counter + seg) : "memory");
    }
    if (LAST && !SPLIT && p.bias != nullptr && nxt < ntiles) {
      long long bm0, bn0;
      tile_origin(nxt, bm0, bn0);
      if (bn0 + BN2 <= p.N) bias_request(bn0);
    }
    if (LAST && act_bwd) aux_request(0, c_buf ^ 1);
    else if (FIRST ? !a1_ahead : (!LAST || r_valid)) request(A1);
    if (PENULT) {
      if (nxt < ntiles) set_cursor(nxt);
      else r_valid = false;
    }
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    phase_barrier();
    multiply(0, 0, fb0);
    phase_barrier();
    // phase 2
    read_b(std::integral_constant<int, 1>{}, fb1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(PENULT || LAST) || r_valid) request(B0);
    phase_barrier();
    multiply(0, 1, fb1);
    phase_barrier();
    // phase 3
    read_a(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    if (!(PENULT || LAST) || r_valid) request(A0);
    phase_barrier();
    multiply(1, 1, fb1);
    phase_barrier();
    // phase 4
    if (!(PENULT || LAST) || r_valid) {
      request(B1);
      if (FIRST && a1_ahead && ahead_stores == 16) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
      else if (FIRST && a1_ahead && ahead_stores == 32) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (!(LAST && act_bwd)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // end of the stream: whatever is still in flight
    }
    c_buf ^= 1;
    frag_addr_flip<A_KMAJOR>(addr_a);
    frag_addr_flip<B_KMAJOR>(addr_b);
    phase_barrier();
    multiply(1, 0, fb0);
    phase_barrier();
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};

  // prologue: all of the first k-step and three half-tiles of the second
  if (!SPLIT && p.bias != nullptr) {
    long long bm0, bn0;
    tile_origin(cur, bm0, bn0);
    if (bn0 + BN2 <= p.N) bias_request(bn0);
  }
  set_cursor(cur);
  request(B0);
  request(A0);
  request(B1);
  request(A1);
  if (n > 1) {
    request(B0);
    request(A0);
    request(B1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    r_valid = false;  // one k-step per tile: one tile per workgroup (the launcher's choice)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  phase_barrier();
  DPA_STAMP(1);

  // The id behind the drawn value in the control word; when this segment is used up (a handful of times per workgroup,
  // at the end of the kernel) the next segments' counters are drawn from, waiting for each answer -- every wave takes
  // that path together, the value it branches on being the workgroup's.
  auto next_from_segments = [&]() __attribute__((always_inline)) -> int {
    unsigned local = static_cast<unsigned>(seg_static) + __builtin_amdgcn_readfirstlane(*(lds_u32_ptr)(smem2 + kCtrlOff));
    while (local >= static_cast<unsigned>(seg_len) && seg_hops < 7) {
      ++seg_hops;
      seg = (seg + 1) & 7;
      seg_range(seg);
      phase_barrier();  // every wave has read the control word
      if (threadIdx.x == 0) {
        const unsigned one = 1u, zero = 0u;
        unsigned got;
        asm volatile("global_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(zero), "v"(one), "s"(tile_counter + seg) : "memory");
        *(lds_u32_ptr)(smem2 + kCtrlOff) = got;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      phase_barrier();
      local = static_cast<unsigned>(seg_static) + __builtin_amdgcn_readfirstlane(*(lds_u32_ptr)(smem2 + kCtrlOff));
    }
    return local < static_cast<unsigned>(seg_len) ? seg_start + static_cast<int>(local) : kNoTile;
  };

  while (true) {
    DPA_STAMP(6);
    DPA_TILE_TIME(0);
    tile_origin(cur, m0, n0);
    // Full interior tiles of a plain product leave through the LDS-staged epilogue below (with the bias
    // already in the accumulators); everything else through the element-wise one.
    const bool staged = !SPLIT && p.beta == 0.f && p.alpha == 1.f && m0 + BM2 <= p.M && n0 + BN2 <= p.N && (p.ldc & 7) == 0 &&
                        ((reinterpret_cast<uintptr_t>(p.C) | reinterpret_cast<uintptr_t>(p.pre_act)) & 15) == 0;
    if (staged && p.bias != nullptr) {
      using lds_u16x4_cptr = const __attribute__((address_space(3))) u16x4*;
      const int bias_lane = fresh_lane();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u16x4 raw = *(lds_u16x4_cptr)(lds_base + kBiasOff + (wn + 16 * j + 4 * (bias_lane >> 4)) * 2);
        f32x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = BF16::to_f32(raw[e]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = b;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // The upper wave row runs one barrier behind the lower one inside the k-steps: on every SIMD one
    // wave multiplies while its partner reads LDS and issues requests.
    if (wr == 1) phase_barrier();
    if (n >= 3) {
      k_step(yes, no, no);
#pragma unroll 1
      for (int t = 0; t < n - 3; ++t) k_step(no, no, no);
      k_step(no, yes, no);
    } else if (n == 2) {
      k_step(yes, yes, no);
    }
    k_step(no, no, yes);
    if (dynamic && wid == 0) {
      // the counter's answer is in (it is older than the requests a vmcnt(6) leaves in flight)
      asm volatile("" : "+v"(fetched));
      *(lds_u32_ptr)(smem2 + kCtrlOff) = __builtin_amdgcn_readfirstlane(fetched);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // written before this wave reaches the barrier below
    }
    if (wr == 0) phase_barrier();
    DPA_STAMP(2);
    DPA_TILE_TIME(2);
    // Every wave is done with the last k-step's buffer: its A1 slot can take the next tile's second k-step
    // now, ahead of this tile's output stores in the memory queue.
    a1_ahead = dynamic && r_valid;
    if (a1_ahead) request(A1);

    if (SPLIT) {
      float* ws = p.workspace + static_cast<long long>(split_idx) * p.M * p.N;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long row = m0 + wm + 16 * i + (lane & 15), col0 = n0 + wn + 16 * j + 4 * (lane >> 4);
          if (row < p.M && col0 < p.N) *reinterpret_cast<f32x4*>(ws + row * p.N + col0) = acc[i][j];  // N % 4 == 0 on this path
        }
      return;
    }
    // Epilogue.  Straight from the accumulators a wave-wide store covers 16 rows x 32 bytes, which the
    // memory system handles at about a third of its streaming rate.  Full interior tiles therefore go
    // through LDS: each wave transposes its 128 x 64 sub-tile 16 rows at a time in a 2 KiB strip of its
    // own (128-byte rows, 16-byte chunks XOR-swizzled by the row: conflict-free 8-byte writes and 16-byte
    // reads) and writes 8 rows x 128 contiguous bytes per store instruction.  Bias and activation are
    // applied in registers first.  The strips lie outside the operand buffers, which already receive the
    // next tile.
    if (staged && act_bwd) {
      // Activation-backward epilogue: C = (alpha*acc) * act'(aux).  Each lane picks the aux values
      // that sit under its accumulators out of the DMA image, writes the result back into the same
      // cells, and the rows then leave as 8 rows x 128 contiguous bytes per store instruction -- so
      // neither direction touches global memory with the accumulator layout's 32-byte row pieces.
      phase_barrier();                     // every wave has finished reading the operand buffers
      const int last = c_buf ^ 1;          // buffer of the last k-step (c_buf points past it); half 0 sits in the other one
      aux_request(1, last);
      u16* Cout = static_cast<u16*>(p.C);
      // Column sums of the bf16-rounded results: each lane adds up its own rows of its 16 columns (16 j + 4 g + e) as it
      // produces them, and the 16 lanes of a row group are summed by rotations at the end -- no LDS traffic (read back
      // from the images the sums were 64 of the 160 LDS instructions per lane and half of the epilogue).
      float cs[4][4] = {};
      const bool want_colsum = p.colsum_partial != nullptr;
      const int el = fresh_lane();
      const int r = el & 15, g = el >> 4;
      // The activation is a compile-time parameter of this code and the column sums are taken whether wanted or not:
      // with a run-time branch per cell the compiler kept the cells apart -- read the cell, wait, look up, wait, write,
      // 64 LDS round trips per lane and tile, 3.2 us per half -- instead of keeping a row of four cells in flight.
      auto process = [&](int half, char* img, auto gelu_c) __attribute__((always_inline)) {
        constexpr bool GELU = decltype(gelu_c)::value;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          char* cell[4];
          u16x4 ax[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            cell[j] = img + (16 * ii + r) * 128 + (((2 * j + (g >> 1)) ^ (r & 7)) << 4) + 8 * (g & 1);
            ax[j] = *reinterpret_cast<const u16x4*>(cell[j]);
          }
          float d[4][4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int bits = ax[j][e];
              if constexpr (GELU) {
                int idx = (bits & 0x7fff) - kGeluLo;
                idx = idx < 0 ? 0 : (idx > kGeluEntries - 1 ? kGeluEntries - 1 : idx);  // non-finite inputs land on 0 / 1
                d[j][e] = gelu_tab[idx + (bits >> 15) * kGeluEntries];
              } else {
                d[j][e] = (bits & 0x8000) || !(bits & 0x7fff) ? 0.f : 1.f;
              }
            }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4& a = acc[4 * half + ii][j];
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[e] = BF16::from_f32(a[e] * d[j][e]);
              cs[j][e] += BF16::to_f32(o[e]);
            }
            *reinterpret_cast<u16x4*>(cell[j]) = o;  // same lane, same cell: read before write by program order
          }
        }
      };
      auto store_rows = [&](int half, const char* img) __attribute__((always_inline)) {
        u32x4 v[8];  // all eight row pieces are read before the first leaves: one LDS round trip instead of eight
#pragma unroll
        for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const u32x4*>(img + (8 * it + (el >> 3)) * 128 + (el & 7) * 16);
        u16* out = Cout + (m0 + wm + 64 * half + (el >> 3)) * p.ldc + n0 + wn + ((el & 7) ^ (el >> 3)) * 8;
#pragma unroll
        for (int it = 0; it < 8; ++it) __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(out + static_cast<long long>(8 * it) * p.ldc));
      };
      char* img0 = smem2 + (last ^ 1) * 2 * kTileBytes2 + wid * 8192;
      char* img1 = smem2 + last * 2 * kTileBytes2 + wid * 8192;
      // a wave only ever touches its own two images, and the DMA counts as complete once its data is in LDS
      auto both_halves = [&](auto gelu_c) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        DPA_TILE_TIME(3);
        process(0, img0, gelu_c);
        DPA_TILE_TIME(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // before any store is in flight: the second half has landed
        DPA_TILE_TIME(5);
        store_rows(0, img0);
        process(1, img1, gelu_c);
        DPA_TILE_TIME(6);
        store_rows(1, img1);
      };
      if (p.aux_mode == kActGelu) both_halves(std::true_type{});
      else both_halves(std::false_type{});
#ifndef GEMM_TILE_TIMES
      if (want_colsum) {
        auto rot = [](float v, auto n_c) __attribute__((always_inline)) {  // v of the lane n places on in its group of 16
          constexpr int ctrl = 0x120 + decltype(n_c)::value;                 // row_ror:n
          return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 out;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = cs[j][e];
            v += rot(v, std::integral_constant<int, 8>{});
            v += rot(v, std::integral_constant<int, 4>{});
            v += rot(v, std::integral_constant<int, 2>{});
            v += rot(v, std::integral_constant<int, 1>{});
            out[e] = v;
          }
          if (r == 0) *reinterpret_cast<f32x4*>(p.colsum_partial + (2 * (m0 / BM2) + (wid >> 2)) * p.N + n0 + wn + 16 * j + 4 * g) = out;
        }
      }
#endif
      DPA_TILE_TIME(1);
      return;
    }
    if (staged) {
      using lds_u16x4_ptr = __attribute__((address_space(3))) u16x4*;
      using lds_u32x4_ptr = const __attribute__((address_space(3))) u32x4*;
      const int l = fresh_lane();  // the epilogue's lane constants are derived here, not kept from the top of the kernel
      const int r = l & 15, g = l >> 4;
      const unsigned mine = lds_base + kStageOff + wid * 2048;
      // + ((chunk ^ (r & 7)) << 4) per column tile; rows 8 .. 15 swap the 8-byte halves of a chunk, so that the 16
      // lanes an 8-byte LDS write serves per cycle (rows 0 .. 15 of one column group) cover all 32 banks
      const unsigned wr_addr = mine + r * 128 + 8 * ((g & 1) ^ (r >> 3));
      const unsigned rd_addr = mine + (l >> 3) * 128 + (l & 7) * 16;
      const int out_col = ((l & 7) ^ (l >> 3)) * 8;
      // One pass per output tensor and 16 rows: the pre-activation (when the backward will need it), then
      // the result.  The activation is a compile-time parameter: selected per element it costs a scalar
      // branch for each of the 128 values a lane owns, more than the arithmetic.  A pass writes its 16 x 64
      // values into the strip (4 x 8 bytes per lane) and reads them back as two 16-byte row pieces; the LDS
      // serves one wave's requests in order, so the passes need no waits between them and are software-
      // pipelined: the next pass is converted and written while the row pieces of the current one return,
      // and only the global stores wait for LDS data.
      auto convert_write = [&](int i, auto act_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4& a = acc[i][j];
          u16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = BF16::from_f32(a[e]);  // alpha == 1 and the bias is in the accumulator on this path
            // the activation sees the pre-activation rounded to bf16, exactly as the unfused path stores and re-reads it
            if (ACT == kActGelu) {
              const int bits = o[e];
              int idx = (bits & 0x7fff) - kGeluLo;
              idx = idx < 0 ? 0 : (idx > kGeluEntries - 1 ? kGeluEntries - 1 : idx);
              o[e] = BF16::from_f32(BF16::to_f32(o[e]) * gelu_tab[idx + (bits >> 15) * kGeluEntries]);
            } else if (ACT != kActNone) {
              o[e] = BF16::from_f32(act_fn(BF16::to_f32(o[e]), ACT));
            }
          }
          *(lds_u16x4_ptr)(wr_addr + (((2 * j + (g >> 1)) ^ (r & 7)) << 4)) = o;
        }
      };
      auto read_rows = [&](u32x4 (&v)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) v[it] = *(lds_u32x4_ptr)(rd_addr + it * 1024);
      };
      auto store_rows = [&](u16* out, int i, const u32x4 (&v)[2]) __attribute__((always_inline)) {
        // rows 8 .. 15 of a pass keep the two 8-byte halves of every 16-byte chunk swapped (see wr_addr)
        // Streaming (non-temporal) stores: the output is far larger than the L2 and nobody reads it before the
        // kernel ends, while the operand panels ARE re-read from L2 by the neighbouring tiles -- written the
        // normal way the output evicts them (measured on 131072 x 2304 x 768: 1124 -> 1241 TFLOP/s, x 768 x 768:
        // 1043 -> 1177; no change at K = 3072 where a tile's stores are a small share of its traffic).
#ifdef GEMM_KO_ALL_STORES  // knock-out (tools/kernel_harness): what do the output stores cost?
        asm volatile("" :: "v"(v[0][0]), "v"(v[0][3]), "v"(v[1][0]), "v"(v[1][3]), "v"(out));
        return;
#endif
        __builtin_nontemporal_store(v[0], reinterpret_cast<u32x4*>(out + static_cast<long long>(16 * i) * p.ldc));
        __builtin_nontemporal_store(u32x4{v[1][2], v[1][3], v[1][0], v[1][1]},
                                    reinterpret_cast<u32x4*>(out + static_cast<long long>(16 * i + 8) * p.ldc));
      };
      auto emit_all = [&](auto act_c) __attribute__((always_inline)) {
        // this lane's element of the tile's first 8 rows
        const long long first = (m0 + wm + (l >> 3)) * p.ldc + n0 + wn + out_col;
        u16* c_rows = static_cast<u16*>(p.C) + first;
        u16* p_rows = p.pre_act ? static_cast<u16*>(p.pre_act) + first : nullptr;
        constexpr std::integral_constant<int, kActNone> plain{};
        u32x4 v[2][2];
        if (p_rows) {
          // passes: (pre-activation, rows i), (result, rows i), i = 0 .. 7
          convert_write(0, plain);
          read_rows(v[0]);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if (q + 1 < 16) {
              if ((q + 1) & 1) convert_write((q + 1) >> 1, act_c);
              else convert_write((q + 1) >> 1, plain);
              read_rows(v[(q + 1) & 1]);
            }
#ifdef GEMM_KO_PRE_STORE  // knock-out (tools/kernel_harness): the pre-activation passes run, their stores do not
            if (q & 1) store_rows(c_rows, q >> 1, v[q & 1]);
            else asm volatile("" :: "v"(v[0][0]), "v"(v[0][1]));
#else
            store_rows(q & 1 ? c_rows : p_rows, q >> 1, v[q & 1]);
#endif
          }
        } else {
          convert_write(0, act_c);
          read_rows(v[0]);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i + 1 < 8) {
              convert_write(i + 1, act_c);
              read_rows(v[(i + 1) & 1]);
            }
            store_rows(c_rows, i, v[i & 1]);
          }
        }
      };
      if (p.act == kActGelu) emit_all(std::integral_constant<int, kActGelu>{});
      else if (p.act == kActRelu) emit_all(std::integral_constant<int, kActRelu>{});
      else emit_all(std::integral_constant<int, kActNone>{});
      ahead_stores = p.pre_act ? 32 : 16;
    } else {
      ahead_stores = 0;  // ragged tile: an unknown number of stores, the next wait covers them all
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          store_tile_bf16(p, acc[i][j], m0 + wm + 16 * i + (lane & 15), n0 + wn + 16 * j + 4 * (lane >> 4));
    }
    DPA_STAMP(3);
    DPA_STAMP_REAL(5);
    DPA_TILE_TIME(1);
#ifdef GEMM_TILE_TIMES
    ++tile_no;
#endif
    if (!dynamic) return;
    // next tile: the id drawn while this one was multiplied becomes the one after it
    cur = nxt;
    if (cur >= ntiles) return;
    nxt = next_from_segments();
    set_offsets(cur);  // the cursor is inside this tile
  }
}

// Sums the split-K partials in index order (deterministic) and applies the regular epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p) {
  const long long quad = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;  // 4 consecutive columns
  const long long nq = p.N >> 2;
  if (quad >= p.M * nq) return;
  const long long row = quad / nq, col0 = (quad % nq) << 2;
  const long long plane = p.M * p.N;
  const float* src = p.workspace + row * p.N + col0;
  f32x4 acc = *reinterpret_cast<const f32x4*>(src);
  for (int s = 1; s < p.split_k; ++s) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + s * plane);
    acc += v;
  }
  store_tile_bf16(p, acc, row, col0);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 16-byte global loads everywhere: contiguous extents, leading dimensions and base addresses in multiples of 8
// elements.  The LDS-DMA kernel and split-K need this; the 128x128 kernel has an element-wise tail path for the rest.
bool gemm_bf16_vector_aligned(const GemmArgs& a) {
  if (a.lda % 8 || a.ldb % 8) return false;
  if (a.a_kmajor ? (a.K % 8) : (a.M % 8)) return false;
  if (a.b_kmajor ? (a.K % 8) : (a.N % 8)) return false;
  if (a.A && !aligned16(a.A)) return false;
  if (a.B && !aligned16(a.B)) return false;
  return true;
}

bool use_large_tile(const GemmArgs& a) {
  if (a.K % BK2 != 0 || !gemm_bf16_vector_aligned(a)) return false;
  if (a.M < 8 || a.N < 8) return false;
  // enough 256x256 tiles to occupy the chip, otherwise the 128x128 kernel spreads better
  const long long tiles = ((a.M + BM2 - 1) / BM2) * ((a.N + BN2 - 1) / BN2);
  return tiles >= 128;
}

}  // namespace

bool gemm_bf16_act_backward_supported(const GemmArgs& a) {
  return gemm_bf16_supported(a) && gemm_bf16_vector_aligned(a) && a.M % BM2 == 0 && a.N % BN2 == 0 && a.K % BK2 == 0 && (a.ldc & 7) == 0 && a.bias == nullptr &&
         a.pre_act == nullptr && a.beta == 0.f && a.alpha == 1.f && a.act == kActNone && aligned16(a.C) && aligned16(a.aux) &&
         (a.aux_mode == kActGelu || a.aux_mode == kActRelu);
}

int gemm_bf16_split_k(const GemmArgs& a) {
  if (!gemm_bf16_supported(a) || !gemm_bf16_vector_aligned(a)) return 1;
  if (a.K % BK2 != 0 || a.N % 4 != 0 || a.M < 8 || a.N < 8) return 1;
  const long long tiles = ((a.M + BM2 - 1) / BM2) * ((a.N + BN2 - 1) / BN2);
  const long long nk = a.K / BK2;
  if (nk < 32) return 1;
  if (tiles >= 128) {
    // Enough tiles to fill the chip, but a tile count just above a multiple of the 256 CUs leaves the
    // last round mostly idle (591 tiles of a [50304, 768] weight gradient: 3 rounds for 2.3 rounds of
    // work).  With a long contraction, s ranges per tile make the rounds s times shorter and the tail
    // with them; taken when that saves at least a tenth and the fp32 partials stay below 1 GiB.
    if (nk < 256) return 1;
    int best = 1;
    double best_cost = static_cast<double>((tiles + 255) / 256);
    const double base = best_cost;
    for (int s = 2; s <= 4; ++s) {
      const double cost = static_cast<double>((tiles * s + 255) / 256) / s;
      if (cost < best_cost - 1e-9 && static_cast<double>(s) * a.M * a.N * 4.0 <= 1073741824.0) {
        best_cost = cost;
        best = s;
      }
    }
    return best_cost <= 0.9 * base ? best : 1;
  }
  // One workgroup per CU (128 KiB of LDS each): aim for one full wave of 256 workgroups, with at
  // least 16 contraction steps per range so the pipeline prologue stays amortised.
  long long s = 256 / tiles;
  if (s > nk / 16) s = nk / 16;
  if (s < 2) return 1;
  const long long iters = (nk + s - 1) / s;
  return static_cast<int>((nk + iters - 1) / iters);
}

bool gemm_bf16_supported(const GemmArgs& a) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
  if (a.lda >= (1ll << 22) || a.ldb >= (1ll << 22)) return false;  // tile-relative byte offsets are 32-bit
  const long long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  return tiles < (1ll << 31);
}

// Tile counters of the streaming (several tiles per workgroup) launches: a small ring of 16-byte slots
// in the code object's own memory, one per launch in flight, zeroed on the launch's stream.
__device__ unsigned g_tile_counters[256 * 8];  // one set of eight (a counter per XCD segment) per launch in flight

struct StreamingSetup {
  unsigned* counters = nullptr;
  int cus = 0;
  unsigned next_slot = 0;
};
StreamingSetup& streaming_setup() {
  static StreamingSetup s = [] {
    StreamingSetup t;
    void* addr = nullptr;
    if (hipGetSymbolAddress(&addr, HIP_SYMBOL(g_tile_counters)) == hipSuccess) t.counters = static_cast<unsigned*>(addr);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) t.cus = prop.multiProcessorCount;
    return t;
  }();
  return s;
}

// Several tiles per workgroup pay off when a tile is short (few k-steps): the cold start and the
// drain of a workgroup are then a visible share of its time.
bool use_streaming(const GemmArgs& a, long long tiles, bool split) {
#ifdef GEMM_NO_STREAMING  // experiment (tools/kernel_harness): one tile per workgroup whatever the shape
  return false;
#endif
  // The activation-backward epilogue needs both LDS buffers: one tile per workgroup.  (Walking the tiles with it as
  // well -- next tile's id drawn during the k-steps, its first two k-steps requested from inside the epilogue as the
  // buffers fall free -- was built and measured: 2152 -> 2120 us standalone on 393216 x 3072 x 768, nothing in the
  // training step; the workgroups then wait for their aux tiles and stores together instead of being relaunched.)
  if (split || a.aux_mode != kActNone) return false;
  const long long nk = a.K / BK2;
  const StreamingSetup& s = streaming_setup();
  return s.counters != nullptr && s.cus > 0 && nk >= 3 && nk <= kStreamingMaxKSteps && tiles >= 3ll * s.cus;
}

bool gemm_bf16_conv_supported(const GemmArgs& a) {
  if (a.conv_mode != 1 && a.conv_mode != 2) return false;
  if (a.conv_B <= 0 || a.conv_H <= 0 || a.conv_W <= 0 || a.conv_C <= 0 || a.conv_C % BK2 != 0) return false;
  const long long halo_pixels = static_cast<long long>(a.conv_B) * (a.conv_H + 2) * (a.conv_W + 2);
  // per-lane byte offsets into the haloed image are 32-bit
  if ((halo_pixels + 2ll * (a.conv_W + 3) + 64) * a.conv_C * 2 >= (1ll << 32)) return false;
  if (a.bias != nullptr || a.pre_act != nullptr || a.aux_mode != kActNone || a.act != kActNone || a.beta != 0.f) return false;
  if (!aligned16(a.A) || !aligned16(a.B)) return false;
  if (a.conv_mode == 1) {
    return a.a_kmajor && a.b_kmajor && a.M == static_cast<long long>(a.conv_B) * a.conv_H * a.conv_W && a.M < (1ll << 31) &&
           a.K == 9ll * a.conv_C && a.ldb == a.K && a.N % 8 == 0 && a.N >= 8;
  }
  return !a.a_kmajor && !a.b_kmajor && a.N == 9ll * a.conv_C && a.ldb == a.conv_C && a.K % BK2 == 0 && a.K >= halo_pixels &&
         a.M % 8 == 0 && a.M >= 8 && a.lda % 8 == 0 && a.workspace != nullptr && a.split_k >= 1;
}

// Fills g_gelu_tables on the current device the first time a launch needs them (on that launch's stream).
void ensure_gelu_tables(hipStream_t stream) {
  static std::atomic<unsigned> done{0};  // bit d: device d is initialised
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned bit = 1u << (dev & 31);
  if (done.load(std::memory_order_acquire) & bit) return;
  hipLaunchKernelGGL(gelu_tables_init_kernel, dim3((2 * kGeluEntries + 255) / 256), dim3(256), 0, stream);
  // Later launches on OTHER streams must see the tables too: wait once.  Inside a stream capture no wait is
  // possible -- the fill then becomes part of the graph (in front of the product that needs it) and the device
  // stays unmarked, so the next launch outside a capture fills and waits.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
  (void)hipStreamSynchronize(stream);
  done.fetch_or(bit, std::memory_order_release);
}

void launch_gemm_bf16(const GemmArgs& a, hipStream_t stream) {
  if (a.act == kActGelu || a.aux_mode == kActGelu) ensure_gelu_tables(stream);
  const bool split = (a.split_k > 1 || a.conv_mode == 2) && a.workspace != nullptr;
  // the activation-backward epilogue exists in the large-tile kernel only (callers check
  // gemm_bf16_act_backward_supported first)
  if (split || use_large_tile(a) || a.aux_mode != kActNone || a.conv_mode != 0) {
    const int tm2 = static_cast<int>((a.M + BM2 - 1) / BM2), tn2 = static_cast<int>((a.N + BN2 - 1) / BN2);
    const long long nk = a.K / BK2;
    const int k_iters = split ? static_cast<int>((nk + a.split_k - 1) / a.split_k) : static_cast<int>(nk);
    const long long tiles = static_cast<long long>(tm2) * tn2;
    unsigned* counter = nullptr;
    unsigned grid_x = static_cast<unsigned>(tiles);
    if (use_streaming(a, tiles, split)) {
      StreamingSetup& s = streaming_setup();
      counter = s.counters + 8 * (s.next_slot++ & 255u);
      (void)hipMemsetAsync(counter, 0, 32, stream);
      grid_x = static_cast<unsigned>(s.cus) & ~7u;  // the same number of workgroups on every XCD
#ifdef GEMM_GRID_OVERRIDE  // experiment (tools/kernel_harness): fewer workgroups than CUs
      grid_x = GEMM_GRID_OVERRIDE;
#endif
    }
    const dim3 grid2(grid_x, split ? static_cast<unsigned>(a.split_k > 1 ? a.split_k : 1) : 1u);
    if (a.conv_mode != 0) {
      static bool conv_configured = false;
      if (!conv_configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_8ph_kernel<true, true, false, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_8ph_kernel<false, false, true, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes2);
        conv_configured = true;
      }
      if (a.conv_mode == 1) {
        hipLaunchKernelGGL((gemm_bf16_8ph_kernel<true, true, false, 1>), grid2, dim3(kThreads2), kLdsBytes2, stream, a, tm2, tn2, k_iters,
                           counter);
      } else {
        hipLaunchKernelGGL((gemm_bf16_8ph_kernel<false, false, true, 2>), grid2, dim3(kThreads2), kLdsBytes2, stream, a, tm2, tn2, k_iters,
                           counter);
        const long long quads = a.M * (a.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, stream, a);
      }
      return;
    }
    // all of a CU's LDS but 960 bytes: above the 64 KiB default, so the limit is raised per kernel
    const size_t lds2 = kLdsBytes2;
    auto go2 = [&](auto ak, auto bk) {
      constexpr bool AK = decltype(ak)::value, BKM = decltype(bk)::value;
      static bool configured = false;
      if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_8ph_kernel<AK, BKM, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_8ph_kernel<AK, BKM, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_8ph_kernel<AK, BKM, false, 0, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes2);
        configured = true;
      }
      if (split) {
        hipLaunchKernelGGL((gemm_bf16_8ph_kernel<AK, BKM, true>), grid2, dim3(kThreads2), lds2, stream, a, tm2, tn2, k_iters, counter);
        const long long quads = a.M * (a.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, stream, a);
      } else if (a.aux_mode != kActNone) {
        hipLaunchKernelGGL((gemm_bf16_8ph_kernel<AK, BKM, false, 0, true>), grid2, dim3(kThreads2), lds2, stream, a, tm2, tn2, k_iters, counter);
      } else {
        hipLaunchKernelGGL((gemm_bf16_8ph_kernel<AK, BKM, false>), grid2, dim3(kThreads2), lds2, stream, a, tm2, tn2, k_iters, counter);
      }
    };
    if (a.a_kmajor && a.b_kmajor) go2(std::true_type{}, std::true_type{});
    else if (a.a_kmajor && !a.b_kmajor) go2(std::true_type{}, std::false_type{});
    else if (!a.a_kmajor && a.b_kmajor) go2(std::false_type{}, std::true_type{});
    else go2(std::false_type{}, std::false_type{});
    return;
  }
  const int tiles_m = static_cast<int>((a.M + BM - 1) / BM), tiles_n = static_cast<int>((a.N + BN - 1) / BN);
  const dim3 grid(static_cast<unsigned>(tiles_m) * static_cast<unsigned>(tiles_n));
  const size_t lds = 4 * kTileBytes;
  const bool vec = gemm_bf16_vector_aligned(a);
  auto go = [&](auto ak, auto bk) {
    constexpr bool AK = decltype(ak)::value, BKM = decltype(bk)::value;
    if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKM, true, BF16>), grid, dim3(kThreads), lds, stream, a, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKM, false, BF16>), grid, dim3(kThreads), lds, stream, a, tiles_m, tiles_n);
  };
  if (a.a_kmajor && a.b_kmajor) go(std::true_type{}, std::true_type{});
  else if (a.a_kmajor && !a.b_kmajor) go(std::true_type{}, std::false_type{});
  else if (!a.a_kmajor && a.b_kmajor) go(std::false_type{}, std::true_type{});
  else go(std::false_type{}, std::false_type{});
}

// fp16 products: the 128x128 kernel with the fp16 MFMA (the LDS-DMA kernel and split-K are bf16 only).
bool gemm_f16_supported(const GemmArgs& a) { return gemm_bf16_supported(a) && a.aux_mode == kActNone; }

void launch_gemm_f16(const GemmArgs& a, hipStream_t stream) {
  const int tiles_m = static_cast<int>((a.M + BM - 1) / BM), tiles_n = static_cast<int>((a.N + BN - 1) / BN);
  const dim3 grid(static_cast<unsigned>(tiles_m) * static_cast<unsigned>(tiles_n));
  const size_t lds = 4 * kTileBytes;
  const bool vec = gemm_bf16_vector_aligned(a);
  auto go = [&](auto ak, auto bk) {
    constexpr bool AK = decltype(ak)::value, BKM = decltype(bk)::value;
    if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKM, true, F16>), grid, dim3(kThreads), lds, stream, a, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKM, false, F16>), grid, dim3(kThreads), lds, stream, a, tiles_m, tiles_n);
  };
  if (a.a_kmajor && a.b_kmajor) go(std::true_type{}, std::true_type{});
  else if (a.a_kmajor && !a.b_kmajor) go(std::true_type{}, std::false_type{});
  else if (!a.a_kmajor && a.b_kmajor) go(std::false_type{}, std::true_type{});
  else go(std::false_type{}, std::false_type{});
}

}  // namespace kernels
}  // namespace dpa
