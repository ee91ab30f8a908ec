// Causal multi-head attention (flash style) for gfx950, head dimension 64, bf16 in / out.
//
// Not an op of the reference (its model is two linear layers, reference min_DDP.py:41-49): it
// exists because the framework's throughput bar is defined on GPT-2, where attention is a
// quarter of the step on the stock kernels.  Everything is computed in the TRANSPOSED
// orientation so that no value ever has to change lanes between the two matrix products:
//
//   S^T[key][q] = K Q^T   v_mfma_f32_16x16x32_bf16, A = K rows (ds_read_b128 from LDS), B = Q
//                         rows (registers).  The accumulator puts the QUERY on the lane
//                         (col = lane & 15) and 4 keys per 16-key tile in registers, so the
//                         softmax statistics of a query are per-lane scalars; the row maximum
//                         needs two 16-lane-stride shuffles per tile, the row sum none until
//                         the very end.
//   O^T[d][q]  += V^T P^T the bf16-packed P^T accumulators are directly the B operand (they are
//                         summed over their register/row index).  The contraction order over
//                         keys is therefore permuted: k-slot (g, j) of a 32-key step is key
//                         32s + 16(j>>2) + 4g + (j&3); the A operand V^T is fetched in exactly
//                         that order by the transposing LDS read ds_read_b64_tr_b16 (4 keys x 16
//                         d per 16-lane group) from the row-major V tile.
//   output                every lane owns 4 consecutive d of one query: 8-byte stores.
//
// q, k, v are read in place from the packed projection output [B, T, 3, H, 64]; the output is
// [B, T, H, 64], which is the layout the output projection consumes.  A workgroup owns 128
// queries (4 waves x 32) of one (batch, head) and walks the key tiles (64 keys) up to the
// diagonal; tiles are double-buffered in LDS with the loads of tile j+1 in flight during the
// math of tile j; workgroups with the most tiles are launched first.
// (That is the first forward kernel and the default backward pair.  Further down: a second forward kernel with 64
// queries per wave on 32x32x16 matrix instructions, and a selectable backward pair on the same instruction.)
#include <cstdlib>

#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

#ifndef FWD32_WAVES_PER_SIMD
#define FWD32_WAVES_PER_SIMD 2
#endif

constexpr int HD = 64;        // head dimension
constexpr int BQ = 128;       // queries per workgroup
constexpr int BKV = 64;       // keys per tile
constexpr int kThreads = 256;
constexpr int kTileBytes = BKV * HD * 2;  // 8 KiB

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// K tile [64 keys][64 d], 128-byte rows, chunk swizzle for row-wise ds_read_b128 fragments.
__device__ __forceinline__ int k_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// V tile [64 keys][64 d], 128-byte rows, swizzled in 32-byte units so the eight 4-key x 16-d
// blocks that a 32-lane half fetches with the transposing read land on disjoint banks.
__device__ __forceinline__ int v_off(int row, int c) { return row * 128 + ((c ^ (2 * ((row >> 1) & 3))) << 4); }

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int key_base, int d_tile, int lane) {
  // 16-lane group g fetches keys key_base + 4g + (0..3) and key_base + 16 + 4g + (0..3) for the 16
  // d values of d_tile; lane i receives d = 16*d_tile + i.
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int c = 2 * d_tile + (pp >> 1), sub = (pp & 1) << 3;
  const int r1 = key_base + 4 * g + qq, r2 = r1 + 16;
  using lds_ptr = __attribute__((address_space(3))) s16x4*;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + v_off(r1, c) + sub));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + v_off(r2, c) + sub));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  bf16x8 r;
  r[0] = static_cast<__bf16>(a0); r[1] = static_cast<__bf16>(a1); r[2] = static_cast<__bf16>(a2); r[3] = static_cast<__bf16>(a3);
  r[4] = static_cast<__bf16>(a4); r[5] = static_cast<__bf16>(a5); r[6] = static_cast<__bf16>(a6); r[7] = static_cast<__bf16>(a7);
  return r;
}


struct AttnParams {
  const u16* q;  // element (b, t, h, d) at q[(b*T + t) * row_stride + h*64 + d]; same for k, v
  const u16* k;
  const u16* v;
  u16* o;        // [B, T, H, 64]
  float* lse;    // [B, H, T], natural log
  long long row_stride;  // elements between consecutive t of q/k/v (3*H*64 for a packed projection)
  int B, H, T;
  float scale_log2;      // softmax scale * log2(e)
  int per_cu;            // resident workgroups per CU of this launch (priority balancing)
};

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Work items.  Under the causal mask, block i of a head (128 queries, or 128 keys in the dK/dV kernel)
// costs i + 1 units (resp. nqb - i), so blocks are handed out in PAIRS (i, nqb-1-i) of equal total cost:
// an item is one pair of one (batch, head), ceil(nqb / 2) items per head (the middle block is alone when
// nqb is odd).  The kernels are persistent: a launch has a fixed number of workgroups (a multiple of 8,
// a few per CU) and workgroup w takes items w, w + G, w + 2G, ... of the XCD-grouped order below, so all
// workgroups carry the same load, nothing is left for a ragged tail, and a workgroup's launch cost is paid
// once per kernel instead of once per block.
// The hardware's issue arbitration favours the OLDER of the workgroups that share a CU: measured on the
// persistent forward kernel at two workgroups per CU, the first-dispatched half of the grid finished its
// (equal) share in 79 us and the second half in 97 us, the last 18 us at half occupancy; raising the
// younger half's priority once simply swaps the roles (96 / 79 us).  So the priority ALTERNATES: in
// every step of its tile loop a workgroup runs at priority 1 when (step + dispatch round) is odd and 0
// otherwise, which gives the CU's resident workgroups the upper hand in turns.  `dispatch_round()` =
// which of the CU's resident workgroups this one is (ids are handed out round by round).
__device__ __forceinline__ int dispatch_round(int per_cu) {
  const int per_round = gridDim.x / (per_cu > 0 ? per_cu : 1);
  return per_round > 0 ? static_cast<int>(blockIdx.x) / per_round : 0;
}
__device__ __forceinline__ void take_turns(int step_plus_round) {
  if (step_plus_round & 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// Item index -> (pair index within the head, batch*head): hardware hands consecutive workgroup ids to the
// 8 XCDs round-robin and every XCD has its own L2, so the items of one head -- which all stream the same
// K / V (or Q / dO) rows -- are numbered such that they land on ONE XCD: ids congruent mod 8 form one
// contiguous run of (head, pair) positions.
__device__ __forceinline__ void item_of_index(int idx, int n_items, int items_per_head, int& pair, int& bh) {
  const int q = n_items >> 3, r = n_items & 7, xcd = idx & 7;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (idx >> 3);
  bh = v / items_per_head;
  pair = v - bh * items_per_head;
}

// Combine a value across the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48) with the
// gfx950 row-swap instructions: two VALU ops per step instead of an LDS round trip (ds_bpermute).
__device__ __forceinline__ float groups_max(float x) {
  const unsigned xi = __float_as_uint(x);
  const u32x2 a = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  const float m1 = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned mi = __float_as_uint(m1);
  const u32x2 b = __builtin_amdgcn_permlane16_swap(mi, mi, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float groups_sum(float x) {
  const unsigned xi = __float_as_uint(x);
  const u32x2 a = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  const float s1 = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned si = __float_as_uint(s1);
  const u32x2 b = __builtin_amdgcn_permlane16_swap(si, si, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// s_waitcnt vmcnt(0) as a real instruction the compiler's wait-count bookkeeping sees: placed after
// the loop-invariant operand loads so that waits inside the loop only cover the loop's own loads
// (the memory counter retires in order; without this every iteration drains the prefetch early).
__device__ __forceinline__ void wait_all_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Raw v_exp_f32 (2^x): inputs here are <= 0 or -inf, where the instruction is exact enough and
// needs none of the denormal-range fix-ups of exp2f().
#ifdef WIDE_KO_EXP  // diagnostic build only: what does the kernel cost without the exponentials?
__device__ __forceinline__ float fast_exp2(float x) { return x; }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

// Waits until at most N of this wave's vector-memory operations (LDS-DMA included) are outstanding;
// emitted as a real s_waitcnt so the compiler's own bookkeeping sees it.
template <int N>
__device__ __forceinline__ void wait_vmem_at_most() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// A [rows][64] bf16 tile image is filled by LDS-DMA (global_load_lds_dwordx4: each lane fetches 16
// bytes and the hardware writes the wave's 64 pieces to 1 KiB of consecutive LDS, i.e. 8 image
// rows).  The destination order is fixed, so the bank swizzle is applied on the SOURCE side: the
// lane that lands in physical chunk pc of row r fetches the logical chunk that belongs there.
// Returns the per-lane source pointer of 1-KiB piece `piece` (rows 8*piece .. 8*piece+7) of a tile
// whose first row is `base` (row stride in elements); V_SWIZZLE selects the v_off image.
template <bool V_SWIZZLE>
__device__ __forceinline__ const u16* dma_src(const u16* base, long long row_stride, int piece, int lane) {
  const int r = piece * 8 + (lane >> 3), pc = lane & 7;
  const int c = V_SWIZZLE ? (pc ^ (2 * ((r >> 1) & 3))) : (pc ^ ((r >> 1) & 7));
  return base + static_cast<long long>(r) * row_stride + c * 8;
}
// Issued through inline assembly on purpose: the compiler orders every LDS read it cannot prove
// disjoint (all transposing reads, for a start) behind ALL outstanding LDS-DMA it knows of with an
// s_waitcnt vmcnt(0), which would serialise the ring.  Ordering is explicit here instead: a counted
// vmcnt wait plus a workgroup barrier before a stage is read, a barrier after its last read before
// it is requested again.  `lds_addr` is the wave-uniform LDS byte address of the 1-KiB piece.
__device__ __forceinline__ void dma_16B(const u16* src, unsigned lds_addr) {
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory");
}
__device__ __forceinline__ unsigned lds_address(const char* p) {
  return __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>((lptr_t)p)));
}

// Output rows of a wave (32 rows x 64 d, bf16) leave through a private LDS strip so that every store
// instruction writes 8 rows x 128 contiguous bytes; straight from the accumulators it would be
// 16 rows x 32 bytes, which the memory system handles at about a third of the rate.  `acc[st][td]`
// holds row 16*st + (lane & 15), d = 16*td + 4*(lane >> 4) + e; `row_scale[st]` multiplies the row.
// The strip (32 rows x 144 bytes) reuses tile memory: the caller has passed a workgroup barrier after
// the last tile read.  Row r of the wave goes to out + r * row_stride (elements).
constexpr int kOutPitch = 144;
constexpr int kOutStripBytes = 32 * kOutPitch;
// `colsum`, when given, receives the 64 column sums of the wave's 32 bf16-rounded rows (lane c writes
// colsum[c]): the piece of a bias gradient that belongs to these rows.
// STREAM: the rows leave through non-temporal stores.  The gradient tensors are read by nobody before the kernel
// ends, while the K / V (Q / dO) rows are re-read from L2 by the head's other blocks; written the normal way the
// gradients evict them (backward 290 -> 277 us at B32; the forward's single output tensor makes no difference
// there, and a small one in the training step: 351.6 -> 350.5 ms).
template <bool STREAM = false>
__device__ __forceinline__ void store_rows_staged(char* strip, const f32x4 (&acc)[2][4], const float (&row_scale)[2], u16* out,
                                                  long long row_stride, int lane, float* colsum = nullptr) {
  const int g = lane >> 4, li = lane & 15;
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int td = 0; td < 4; ++td) {
      u16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = BF16::from_f32(acc[st][td][e] * row_scale[st]);
      *reinterpret_cast<u16x4*>(strip + (16 * st + li) * kOutPitch + (16 * td + 4 * g) * 2) = w;
    }
  // a wave reads back only its own writes; the LDS serves one wave's requests in order
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = 8 * it + (lane >> 3), ch = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(strip + r * kOutPitch + ch * 16);
    if (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + r * row_stride + ch * 8));
    else *reinterpret_cast<u32x4*>(out + r * row_stride + ch * 8) = v;
  }
  if (colsum) {
    float cs = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) cs += BF16::to_f32(*reinterpret_cast<const u16*>(strip + r * kOutPitch + lane * 2));
    colsum[lane] = cs;
  }
}

// Column sums of a workgroup's 128 output rows (bias gradient pieces): every wave has left the sums of its 32 rows in
// `red[wid][64]` (LDS, through the `colsum` argument of the row stores); after a barrier the first wave adds the four
// and writes one row of the partial-sum tensor -- a quarter of the rows (and of the traffic of the reduction that
// follows) that one row per wave would need.  `n_sets`: how many [4][64] sets were filled (dK and dV share a barrier).
constexpr int kColsumSetFloats = 4 * 64;
__device__ __forceinline__ void emit_block_colsums(const float* red, int n_sets, float* const (&dst)[2], int lane, int wid) {
  __syncthreads();
  if (wid == 0) {
    for (int s = 0; s < n_sets; ++s) {
      const float* r = red + s * kColsumSetFloats;
      dst[s][lane] = (r[lane] + r[64 + lane]) + (r[128 + lane] + r[192 + lane]);
    }
  }
}

// Online-softmax update of one 16-query strip: scores s (S^T accumulators of the strip's four
// 16-key tiles) -> bf16 probabilities packed as the two B operands of the P.V product.  The loop is
// bound by vector issue slots, not by the matrix pipe, so the work per score is one v_fma_f32 and one
// v_exp_f32 (plain, not packed, instructions: packed f32 VALU ops issue at less than half the rate of
// two plain ones next to matrix instructions):
//  * the running maximum m is kept in the scaled (log2) domain but taken over the RAW scores, so the
//    softmax scale rides in the exponent's FMA: p = exp2(s * c - m);
//  * m only moves when a score exceeds it by more than kDeferLog2 (probabilities stay below
//    2^kDeferLog2, harmless in fp32 sums and bf16 operands).  The common case is decided per lane -- if
//    no lane of the wave holds a score above the threshold, no query's maximum moved -- so the
//    cross-lane maximum and the rescale of the output only run in the rare tile where something moved;
//  * the row sum is not added up on the VALU at all: see fwd_tile.
// (Measured and dropped: pre-scaled Q with -m as the score product's initial accumulator, which would
// leave only the v_exp_f32 -- the compiler copies the initial value into the result registers first,
// 16 moves per strip, and the longer live ranges cost a wave of occupancy.)
constexpr float kDeferLog2 = 8.f;

template <bool MASK>
__device__ __forceinline__ void softmax_strip(f32x4 (&s)[4], f32x4 (&o)[4], f32x4& lsum, float& m, bf16x8 (&pf)[2], int rel, float c) {
  if (MASK) {  // key index 16t + e of this lane group is allowed while 16t + e <= rel
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (16 * t + e > rel) s[t][e] = -__builtin_huge_valf();
  }
  float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
  mx = fmaxf(fmaxf(mx, s[0][3]), s[1][0]);
  mx = fmaxf(fmaxf(mx, s[1][1]), s[1][2]);
  mx = fmaxf(fmaxf(mx, s[1][3]), s[2][0]);
  mx = fmaxf(fmaxf(mx, s[2][1]), s[2][2]);
  mx = fmaxf(fmaxf(mx, s[2][3]), s[3][0]);
  mx = fmaxf(fmaxf(mx, s[3][1]), s[3][2]);
  mx = fmaxf(mx, s[3][3]) * c;          // c > 0: this lane's maximum of the scaled scores
  if (__any(mx > m + kDeferLog2)) {     // rare after the first tile: some query's maximum moved a lot
    const float mg = groups_max(mx);    // the query's maximum over all four lane groups
    const float m_new = mg > m + kDeferLog2 ? mg : m;
    const float alpha = fast_exp2(m - m_new);
    m = m_new;
#pragma unroll
    for (int td = 0; td < 4; ++td)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[td][e] *= alpha;
#pragma unroll
    for (int e = 0; e < 4; ++e) lsum[e] *= alpha;
  }
  const float nm = -m;
  float pv[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) pv[t][e] = fast_exp2(fmaf(s[t][e], c, nm));
#pragma unroll
  for (int sk = 0; sk < 2; ++sk)
    pf[sk] = pack8(pv[2 * sk][0], pv[2 * sk][1], pv[2 * sk][2], pv[2 * sk][3], pv[2 * sk + 1][0], pv[2 * sk + 1][1],
                   pv[2 * sk + 1][2], pv[2 * sk + 1][3]);
}

// One key tile for both 16-query strips of a wave.  MASK = the tile crosses the diagonal.
// (Issuing both strips' score products before the first softmax, so that the second strip's matrix work
// runs under the first strip's VALU work, was measured twice -- at three and at two waves per SIMD --
// and is within run-to-run noise both times.)
// The row sums of P come out of the matrix pipe: lsum[st] += 1^T P^T, one more 16x16x32 product per
// strip and 32 keys with an all-ones A operand (every accumulator row then holds the sum for the
// lane's query, of exactly the bf16 values that multiply V).  That replaces 32 VALU additions per
// strip and tile by 2 matrix instructions.
template <bool MASK>
__device__ __forceinline__ void fwd_tile(const char* ks, const char* vs, const bf16x8 (&qf)[2][2], f32x4 (&o)[2][4], float (&m)[2],
                                         f32x4 (&lsum)[2], int key0, int q0, int lane, float c) {
  const int g = lane >> 4, li = lane & 15;
  // K fragments (A operand): row = key 16t + li, d chunk 4*kk + g
  bf16x8 kf[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) kf[t][kk] = *reinterpret_cast<const bf16x8*>(ks + k_off(16 * t + li, 4 * kk + g));
  const int rel0 = q0 + li - key0 - 4 * g;
  bf16x8 pf[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][kk], qf[st][kk], s[t], 0, 0, 0);
    }
    softmax_strip<MASK>(s, o[st], lsum[st], m[st], pf[st], rel0 + 16 * st, c);
  }
  const s16x8 ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_bits);
#pragma unroll
  for (int sk = 0; sk < 2; ++sk)
#pragma unroll
    for (int st = 0; st < 2; ++st) lsum[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[st][sk], lsum[st], 0, 0, 0);
  // O^T += V^T P^T: V fragments are shared by both strips
#pragma unroll
  for (int td = 0; td < 4; ++td)
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
      const bf16x8 vf = tr_frag(vs, 32 * sk, td, lane);
#pragma unroll
      for (int st = 0; st < 2; ++st) o[st][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[st][sk], o[st][td], 0, 0, 0);
    }
}

#ifdef ATTN_STAMP
__device__ unsigned long long g_attn_stamps[1 << 12][8];  // diagnostic build: per workgroup {loop cycles, tiles, rt entry, rt exit, life cycles, sync cycles}
#define STAMP_ARGS , unsigned long long& st_loop, unsigned long long& st_tiles, unsigned long long& st_sync
#else
#define STAMP_ARGS
#endif
// One block of 128 queries of one (batch, head): the whole former kernel body.
__device__ __forceinline__ void attn_fwd_block(const AttnParams& p, char* smem, int bh, int qb, int round STAMP_ARGS) {
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int q0 = qb * BQ + wid * 32;

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const u16* K = p.k + head_off;
  const u16* V = p.v + head_off;

  f32x4 o[2][4], lsum[2];
  float m[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    m[st] = -__builtin_huge_valf();
    lsum[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int td = 0; td < 4; ++td) o[st][td] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // K / V tiles travel global -> LDS by DMA through a ring of three stages: the tile two steps
  // ahead is requested right after the barrier that publishes the current one, so every request
  // has a whole tile of math to land behind.  Waves 0-1 move K pieces, waves 2-3 V pieces: four
  // 1-KiB pieces per wave and tile.
  const int n_tiles = (qb * BQ + BQ) / BKV;
  const bool moves_v = wid >= 2;
  const int piece0 = (wid & 1) * 4;
  const u16* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    src[i] = moves_v ? dma_src<true>(V, p.row_stride, piece0 + i, lane) : dma_src<false>(K, p.row_stride, piece0 + i, lane);
  const long long tile_step = static_cast<long long>(BKV) * p.row_stride;
  const int dst0 = (moves_v ? kTileBytes : 0) + piece0 * 1024;
  const unsigned smem_addr = lds_address(smem);
  auto request = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma_16B(src[i], smem_addr + stage * 2 * kTileBytes + dst0 + i * 1024);
      src[i] += tile_step;
    }
  };
  __syncthreads();  // the previous block's output strips (same memory) have been read back by every wave
  request(0);
  if (n_tiles > 1) request(1);
  // Q fragments (B operand): lane = query li of strip st, d = 32*kk + 8*g .. +7.  Loaded AFTER the first
  // tiles were requested, so that a block pays one trip to memory before its first product, not two.
  bf16x8 qf[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qf[st][kk] = *reinterpret_cast<const bf16x8*>(Q + static_cast<long long>(q0 + 16 * st + li) * p.row_stride + 32 * kk + 8 * g);
  wait_all_vmem();
#ifdef ATTN_STAMP
  const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif

  int stage = 0;
  for (int jt = 0; jt < n_tiles; ++jt) {
#ifdef ATTN_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    take_turns(jt + round);
    if (jt + 1 < n_tiles) wait_vmem_at_most<4>();  // this wave's pieces of tile jt have landed
    else wait_vmem_at_most<0>();
    __syncthreads();  // every wave's pieces are visible; nobody still reads the stage requested next
#ifdef ATTN_STAMP
    __builtin_amdgcn_sched_barrier(0);
    st_sync += __builtin_amdgcn_s_memtime() - ta;
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (jt + 2 < n_tiles) request(stage == 0 ? 2 : stage - 1);
    const char* ks = smem + stage * 2 * kTileBytes;
    const char* vs = ks + kTileBytes;
    const int key0 = jt * BKV;
    if (key0 <= q0 + 31) {  // otherwise every key of the tile is in this wave's future
      if (key0 + BKV - 1 > q0) fwd_tile<true>(ks, vs, qf, o, m, lsum, key0, q0, lane, p.scale_log2);
      else fwd_tile<false>(ks, vs, qf, o, m, lsum, key0, q0, lane, p.scale_log2);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
#ifdef ATTN_STAMP
  st_loop += __builtin_amdgcn_s_memtime() - t_loop0;
  st_tiles += n_tiles;
#endif

  // Normalise and store.  Lane: query li of strip st, d = 16*td + 4*g + e.
  float inv[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const float lt = lsum[st][0];  // every accumulator row holds the full sum over keys for this lane's query
    inv[st] = 1.f / lt;
    const int qi = q0 + 16 * st + li;
    if (g == 0) p.lse[(static_cast<long long>(b) * p.H + h) * p.T + qi] = (m[st] + log2f(lt)) * 0.6931471805599453f;
  }
  __syncthreads();  // all waves are done with the tile stages: their memory becomes the output strips
  const long long o_stride = static_cast<long long>(p.H) * HD;
  store_rows_staged<true>(smem + wid * kOutStripBytes, o, inv, p.o + (static_cast<long long>(b) * p.T + q0) * o_stride + static_cast<long long>(h) * HD,
                    o_stride, lane);
}

// Two workgroups per CU: forcing three (168 registers) made the kernel 11 % slower.
__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[6 * kTileBytes];  // [3 stages][K | V]
  const int round = dispatch_round(p.per_cu);
  const int nqb = p.T / BQ, per_head = (nqb + 1) >> 1, n_items = per_head * p.B * p.H;
#ifdef ATTN_STAMP
  unsigned long long st_loop = 0, st_tiles = 0, st_sync = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
#define STAMP_PASS , st_loop, st_tiles, st_sync
#else
#define STAMP_PASS
#endif
  for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    int pair, bh;
    item_of_index(idx, n_items, per_head, pair, bh);
    const int hi = nqb - 1 - pair;
    for (int half = 0; half < (pair != hi ? 2 : 1); ++half)  // the long block of the pair, then its short partner
      attn_fwd_block(p, smem, bh, half == 0 ? hi : pair, round STAMP_PASS);
  }
#ifdef ATTN_STAMP
  if (threadIdx.x == 64 * 3 && blockIdx.x < (1 << 12)) {
    unsigned long long* r = g_attn_stamps[blockIdx.x];
    r[0] = st_loop; r[1] = st_tiles; r[2] = rt0; r[3] = __builtin_amdgcn_s_memrealtime(); r[4] = __builtin_amdgcn_s_memtime() - c0; r[5] = st_sync;
  }
#endif
}

// =====================================================================================
// Forward, 64 queries per wave on 32x32x16 matrix instructions ("wide" variant; launch_attention_fwd chooses
// between the two forward kernels by shape).
//
// The kernel above leaves all overlap of matrix work and softmax to the hardware's switching between its two
// waves per SIMD; measured, a SIMD then spends ~1460 cycles per 2048 (query, key) pairs although the
// matrix pipe needs 576 and the vector ALU ~640.  Here a wave owns 64 queries = two query blocks of 32
// (the accumulator column = lane of v_mfma_f32_32x32x16_bf16) and overlaps the stages itself: the two query
// blocks run half a step out of phase, so that each half-step pairs the softmax of one query block (vector
// ALU) with the matrix work of the other, and the two never depend on each other:
//     half 0 of step b:  P(b)[0] = exp2(S(b)[0] c - m[0])   |  S(b)[1]   = K(b)   Q[1]^T ,  O[1] += V(b-1)^T P(b-1)[1]
//     half 1 of step b:  P(b)[1] = exp2(S(b)[1] c - m[1])   |  S(b+1)[0] = K(b+1) Q[0]^T ,  O[0] += V(b)^T   P(b)[0]
// (b = 32-key block).  Every value has one home: one score buffer and one probability buffer per query block,
// one set of K and of V fragments, which are re-read from LDS right behind the last matrix instruction that
// used the old ones -- four matrix instructions (128 cycles) before the first one that needs the new ones.
// The running maximum m of a query only moves when a score exceeds it by more than kDeferLog2; the test runs
// on a query block's fresh scores between two halves, when the probabilities of that query block have just been
// consumed, so in the rare case it fires only the output accumulators and the row sum are rescaled.
// Row sums: v_dot2c_f32_bf16 of the packed bf16 probability pairs with (1, 1): 16 instructions per block and
// wave instead of 32 additions, and exactly the rounded values that multiply V.
// A workgroup is 4 waves, one on each SIMD (the hardware spreads the waves of ONE workgroup over the SIMDs; two
// 2-wave workgroups per CU were measured to leave a SIMD empty and another with two waves on 93 % of the CUs).
// Its two wave pairs are independent: each works on 128 queries of its own head -- the same query blocks of two
// neighbouring heads, so both pairs take the same number of steps and can share the workgroup barrier -- with
// its own LDS.  K / V blocks (4 KiB each) go global -> LDS by DMA through two rings of four blocks per pair,
// requested two steps ahead.  Two workgroups per CU: measured with one (one wave per SIMD), a wave spends
// more than half of every step in places a second wave can fill -- the counted DMA wait and the barrier, the DMA
// requests themselves (~60 cycles each), the code between the halves -- and the vector ALU port that matrix and
// vector instructions of a wave share leaves little of the former to hide under the latter anyway (8 matrix + 48
// vector instructions: 264 + ~290 cycles apart, 460-520 together, tools/kernel_harness/mfma_chain_probe.hip).
// =====================================================================================
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int kWideThreads = 256;
constexpr int KB = 32;                                  // keys per step
constexpr int kBlkBytes = KB * HD * 2;                  // 4 KiB
constexpr int kRing = 4;                                // blocks per ring (K and V each)
constexpr int kAhead = 2;                               // a block is requested this many steps before the step that reads it
constexpr int kWideStripBytes = 64 * kOutPitch;         // one wave's 64 output rows
constexpr int kWidePairLds = 2 * kRing * kBlkBytes;     // one wave pair: its two rings; the output strips reuse them
static_assert(2 * kWideStripBytes <= kWidePairLds, "both waves' output strips must fit into the pair's rings");
constexpr int kWideLdsBytes = 2 * kWidePairLds;

// V block [32 keys][64 d], 128-byte rows: the transposing read of a 32-lane half fetches 4 keys x 32 d
// (64 bytes of 4 consecutive rows); rows r and r + 2 share banks unless their 64-byte halves are swapped.
__device__ __forceinline__ int v2_off(int row, int c) { return row * 128 + ((c ^ (((row >> 1) & 1) << 2)) << 4); }

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
}
// 1 KiB (8 rows) global -> LDS: lane fetches 16 bytes at base + off, the wave's pieces land at lds_dst + 16 * lane.
__device__ __forceinline__ void dma_piece(const char* base, unsigned off, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ float half_swap_max(float x) {   // max with the lane 32 away
  const unsigned xi = __float_as_uint(x);
  const u32x2 a = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
  const unsigned xi = __float_as_uint(x);
  const u32x2 a = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

// K fragments of one block (A operand of the score product): lane = key (lane & 31), d = 16*ds + 8*(lane >> 5) ..
// (k_off(row, c) for c = 2*ds + hi differs from ds = 0 only in address bits 5-6: one lane constant, XOR per ds.)
__device__ __forceinline__ void wide_load_k(const char* ring, int blk_plus_koff, bf16x8 (&kf)[4]) {
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) kf[ds] = *reinterpret_cast<const bf16x8*>(ring + (blk_plus_koff ^ (ds << 5)));
}
// V^T fragments of one block (A operand of the output product): lane = d (32*db + (lane & 31)); the eight k-slots
// of lane half hi in 16-key step s are keys 16s + 4hi + (0..3) and 16s + 8 + 4hi + (0..3): the keys the score
// accumulator keeps in registers 8s .. 8s+7 of that half.
// (db = 1 is 4 chunks further, which the swizzle turns into a flip of address bit 6.)
__device__ __forceinline__ void wide_load_v(const char* ring, int blk_plus_voff, bf16x8 (&vf)[2][2]) {
  using lds_ptr = __attribute__((address_space(3))) s16x4*;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char* a = ring + (blk_plus_voff ^ (db << 6));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + s * 2048));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + s * 2048 + 1024));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      vf[db][s] = __builtin_bit_cast(bf16x8, v);
    }
}

// One half-step (see the header of this section): the softmax of query block Q's scores s[Q] -> pr[Q], under
// the matrix work of the other query block R: fresh scores s[R] (DO_S) and its output product with pr[R] (DO_O).
// Half 0 re-reads the K fragments (block address k_addr) behind the score product, half 1 the V fragments
// (v_addr) in front of it.
template <int Q, bool DO_S, bool DO_O>
__device__ __forceinline__ void wide_half(f32x16 (&s)[2], bf16x8 (&pr)[2][2], bf16x8 (&kf)[4], bf16x8 (&vf)[2][2], const bf16x8 (&qf)[2][4],
                                          f32x16 (&o)[2][2], const float (&m)[2], float (&l)[2], float c, const char* kring,
                                          const char* vring, int k_addr, int v_addr) {
  constexpr int R = 1 - Q;
  // Fragment reads sit four matrix instructions (128 cycles) ahead of their first use: half 1 fetches V(b) in front
  // of its score product, half 0 K(b+1) behind its score product, which was the last user of K(b).  (Dependent
  // matrix instructions that accumulate into the same registers issue back to back without a stall.)
#ifdef WIDE_KO_MFMA  // diagnostic build only: what does the kernel cost without the matrix instructions?
  constexpr bool kMatrix = false;
#else
  constexpr bool kMatrix = true;
#endif
  if (Q == 1) wide_load_v(vring, v_addr, vf);
  if (DO_S && kMatrix) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s[R][e] = 0.f;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) s[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds], qf[R][ds], s[R], 0, 0, 0);
  }
  if (Q == 0) wide_load_k(kring, k_addr, kf);
  if (DO_O && kMatrix) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) o[R][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][s2], pr[R][s2], o[R][db], 0, 0, 0);
  }
  if (!kMatrix) {  // keep the fragment reads alive
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : : "v"(kf[i]), "v"(vf[i & 1][i >> 1]));
  }
  bf16x2 ones;
  ones[0] = static_cast<__bf16>(1.f);
  ones[1] = static_cast<__bf16>(1.f);
  const float nm = -m[Q];
  float lp = l[Q];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x2 v;
      v[0] = static_cast<__bf16>(fast_exp2(fmaf(s[Q][8 * s2 + 2 * j], c, nm)));
      v[1] = static_cast<__bf16>(fast_exp2(fmaf(s[Q][8 * s2 + 2 * j + 1], c, nm)));
      lp = __builtin_amdgcn_fdot2_f32_bf16(v, ones, lp, false);
      w[j] = __builtin_bit_cast(unsigned, v);
    }
    const u32x4 packed = {w[0], w[1], w[2], w[3]};
    pr[Q][s2] = __builtin_bit_cast(bf16x8, packed);
  }
  l[Q] = lp;
  // 1 matrix instruction : 6 vector instructions (48 per half: 16 fma, 16 exp, 8 convert, 8 dot); LDS reads where they stand
  if (Q == 1) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
  if (DO_S) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, (DO_O ? 6 : 12), 0);
    }
  }
  if (Q == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
  if (DO_O) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, (DO_S ? 6 : 12), 0);
    }
  }
  // The probabilities are only consumed in the next half; without a use here the compiler sinks the softmax out of
  // this block, past the branches that follow, where no matrix instruction is left to hide it.
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) asm volatile("" : : "v"(pr[Q][s2]));
  asm volatile("" : : "v"(l[Q]));
}

// Between two halves: the scores s[R] of block j are complete and pr[R] has been consumed.  Applies the causal mask
// when the block touches the diagonal (rel = query - first key of this lane half; register i holds key
// (i & 3) + 8 * (i >> 2) of it) and moves the running maximum when some score is more than kDeferLog2 above it.
template <int R>
__device__ __forceinline__ void wide_between(f32x16 (&s)[2], float (&m)[2], float (&l)[2], f32x16 (&o)[2][2], bool mask, int rel, float c) {
  if (mask) {
    asm volatile("" ::: "memory");  // a real (wave-uniform) branch: folded into selects, the compare / select pairs would run in every step
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((i & 3) + 8 * (i >> 2) > rel) s[R][i] = -__builtin_huge_valf();
  }
  float t = fmaxf(fmaxf(s[R][0], s[R][1]), s[R][2]);
#pragma unroll
  for (int i = 3; i < 15; i += 2) t = fmaxf(fmaxf(t, s[R][i]), s[R][i + 1]);
  const float mx = fmaxf(t, s[R][15]) * c;
  if (__any(mx > m[R] + kDeferLog2)) {
    const float mg = half_swap_max(mx);  // both lanes of a query must agree
    const float m_new = mg > m[R] + kDeferLog2 ? mg : m[R];
    const float alpha = m_new == m[R] ? 1.f : fast_exp2(m[R] - m_new);
    m[R] = m_new;
    l[R] *= alpha;
    // in place, as assembly: a plain multiply lets the register allocator give the products new homes, and every
    // pass that does NOT take this branch then pays the copies that reconcile the two
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(o[R][db][e]) : "v"(alpha));
  }
}

// Last output product of query block 1, normalisation, log-sum-exp and the wave's 64 output rows (through its LDS strip).
__device__ __forceinline__ void wide_finish(const AttnParams& p, char* strip, f32x16 (&o)[2][2], const bf16x8 (&pr)[2][2],
                                            const bf16x8 (&vf)[2][2], const float (&m)[2], const float (&l)[2], int b, int h, int qw,
                                            int lane) {
  const int hi = lane >> 5, lq = lane & 31;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int db = 0; db < 2; ++db) o[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][s2], pr[1][s2], o[1][db], 0, 0, 0);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float lt = half_swap_sum(l[qb]);
    const float inv = 1.f / lt;
    if (hi == 0) p.lse[(static_cast<long long>(b) * p.H + h) * p.T + qw + 32 * qb + lq] = (m[qb] + log2f(lt)) * 0.6931471805599453f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = BF16::from_f32(o[qb][db][4 * j + e] * inv);
        *reinterpret_cast<u16x4*>(strip + (32 * qb + lq) * kOutPitch + (32 * db + 8 * j + 4 * hi) * 2) = w;
      }
  }
  // a wave reads back only its own writes; the LDS serves one wave's requests in order
  const long long o_stride = static_cast<long long>(p.H) * HD;
  u16* out = p.o + (static_cast<long long>(b) * p.T + qw) * o_stride + static_cast<long long>(h) * HD;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = 8 * it + (lane >> 3), ch = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(strip + r * kOutPitch + ch * 16);
    *reinterpret_cast<u32x4*>(out + r * o_stride + ch * 8) = v;
  }
}

#ifdef WIDE_STAMP
// Diagnostic build (tools/kernel_harness/attn_wide_stamps.hip): per wave, cycles spent in the six sections of a step.
__device__ unsigned long long g_wide_stamps[1024][20];
#define WIDE_T(var)                                                         \
  unsigned long long var;                                                   \
  __builtin_amdgcn_sched_barrier(0);                                        \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
  __builtin_amdgcn_sched_barrier(0)
#define WIDE_ACC(i, a, b) wide_acc[i] += (b) - (a)
#define WIDE_STAMP_ARG , unsigned long long (&wide_acc)[20]
#define WIDE_STAMP_PASS , wide_acc
#else
#define WIDE_T(var)
#define WIDE_ACC(i, a, b)
#define WIDE_STAMP_ARG
#define WIDE_STAMP_PASS
#endif
// One block of 128 queries of one (batch, head), by one wave pair (`smem` is the pair's own LDS).
__device__ __forceinline__ void attn_fwd_wide_block(const AttnParams& p, char* smem, int bh, int qblk WIDE_STAMP_ARG) {
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 1);
  const int hi = lane >> 5, lq = lane & 31;
  const int q0 = qblk * BQ, qw = q0 + wid * 64;
  const int nb_wg = (q0 + BQ) / KB, nb_w = (qw + 64) / KB;  // key blocks of the workgroup / of this wave (>= 2)
  const int last = nb_w - 1;

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const char* Kb = reinterpret_cast<const char*>(p.k + head_off);
  const char* Vb = reinterpret_cast<const char*>(p.v + head_off);
  const long long blk_step = static_cast<long long>(KB) * p.row_stride * 2;  // bytes between consecutive key blocks

  char* kring = smem;
  char* vring = smem + kRing * kBlkBytes;
  char* strip = smem + wid * kWideStripBytes;  // inside the rings: written after the pair's last step
  const unsigned kring_addr = lds_address(kring), vring_addr = lds_address(vring);

  // lane constants of the fragment reads (ds = 0, db = 0; see wide_load_k / wide_load_v)
  const int koff = k_off(lq, hi);
  int voff;
  {
    const int i = lane & 15, g1 = (lane >> 4) & 1, pp = i & 3, row = 4 * hi + (i >> 2);
    voff = v2_off(row, 2 * g1 + (pp >> 1)) + ((pp & 1) << 3);
  }
  // This wave moves pieces wid and wid + 2 (8 rows each) of every K block and every V block: rows 16 apart have
  // the same swizzle, so the second piece is the first at a fixed distance (added to the scalar base).
  unsigned k_src, v_src;
  {
    const int r = wid * 8 + (lane >> 3), pc = lane & 7;
    k_src = static_cast<unsigned>((static_cast<long long>(r) * p.row_stride + (pc ^ ((r >> 1) & 7)) * 8) * 2);
    v_src = static_cast<unsigned>((static_cast<long long>(r) * p.row_stride + (pc ^ (((r >> 1) & 1) << 2)) * 8) * 2);
  }
  const long long piece_step = 16 * p.row_stride * 2;
  auto request_k = [&](int j) {
    const char* base = uniform_ptr(Kb + j * blk_step);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma_piece(base + i * piece_step, k_src, kring_addr + (j & (kRing - 1)) * kBlkBytes + (wid + 2 * i) * 1024);
  };
  auto request_v = [&](int j) {
    const char* base = uniform_ptr(Vb + j * blk_step);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma_piece(base + i * piece_step, v_src, vring_addr + (j & (kRing - 1)) * kBlkBytes + (wid + 2 * i) * 1024);
  };

  // Requests are issued as if by steps -kAhead .. -1 of the loop below (step t asks for V(t + kAhead) and
  // K(t + kAhead + 1)), after K(0); the Q loads ride behind the first three, which the first wait covers.
  WIDE_T(tb0);
  __syncthreads();  // the previous item's last reads of the rings and of the strip are done
  request_k(0);
  request_v(0);
  request_k(1);
  // Q fragments (B operand of the score product): lane = query 32*qb + lq, d = 16*ds + 8*hi ..
  bf16x8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
      qf[qb][ds] = *reinterpret_cast<const bf16x8*>(Q + static_cast<long long>(qw + 32 * qb + lq) * p.row_stride + 16 * ds + 8 * hi);
#pragma unroll
  for (int t = 1 - kAhead; t < 0; ++t) {
    if (t + kAhead < nb_wg) request_v(t + kAhead);
    if (t + kAhead + 1 < nb_wg) request_k(t + kAhead + 1);
  }

  f32x16 o[2][2], s[2];
  bf16x8 pr[2][2], kf[4], vf[2][2];
  float m[2], l[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m[qb] = -__builtin_huge_valf();
    l[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][db][e] = 0.f;
  }
  // Outstanding requests of this wave that a step may leave in flight: those of the kAhead - 1 steps before it.
  auto younger = [&](int bb) {
    const int v_hi = min(nb_wg, bb + kAhead), k_hi = min(nb_wg, bb + kAhead + 1);
    return 2 * max(0, v_hi - (bb + 1)) + 2 * max(0, k_hi - (bb + 2));
  };
  auto wait_landed = [&](int bb) {  // K(bb+1) and V(bb) of this wave's pieces have landed
    switch (younger(bb)) {
      case 4: wait_vmem_at_most<4>(); break;
      case 2: wait_vmem_at_most<2>(); break;
      default: wait_vmem_at_most<0>(); break;
    }
  };
  static_assert(kAhead == 2, "the wait table above lists the counts for a lead of two steps");
  wait_landed(0);
  __syncthreads();

  const float c = p.scale_log2;
  // rel of query block R and key block j = query - first key of the lane half; block j touches the diagonal of
  // query block 0 from nb_w - 2 on, of query block 1 only as the last one.
  const int rel_q0 = qw + lq - 4 * hi;
  auto rel = [&](int r, int j) { return rel_q0 + 32 * r - 32 * j; };
  auto k_addr = [&](int j) { return ((j & (kRing - 1)) * kBlkBytes) | koff; };
  auto v_addr = [&](int j) { return ((j & (kRing - 1)) * kBlkBytes) + voff; };

  // S(0)[0]
  wide_load_k(kring, koff, kf);
#pragma unroll
  for (int e = 0; e < 16; ++e) s[0][e] = 0.f;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds], qf[0][ds], s[0], 0, 0, 0);
  wide_between<0>(s, m, l, o, 0 >= last - 1, rel(0, 0), c);

  auto step_sync = [&](int bb) {  // K(bb+1) and V(bb) have landed and are visible; the blocks read in step bb-2 are free
    WIDE_T(ta);
    if (bb > 0) {
      wait_landed(bb);
      WIDE_T(tb);
      WIDE_ACC(0, ta, tb);
      __syncthreads();
      WIDE_T(tc);
      WIDE_ACC(1, tb, tc);
    }
    WIDE_T(td);
    if (bb + kAhead < nb_wg) request_v(bb + kAhead);
    if (bb + kAhead + 1 < nb_wg) request_k(bb + kAhead + 1);
    WIDE_T(te);
    WIDE_ACC(2, td, te);
  };

  WIDE_T(tb1);
  WIDE_ACC(8, tb0, tb1);
  // step 0: no output product of query block 1 yet
  step_sync(0);
  wide_half<0, true, false>(s, pr, kf, vf, qf, o, m, l, c, kring, vring, k_addr(1), 0);
  wide_between<1>(s, m, l, o, 0 >= last, rel(1, 0), c);
  wide_half<1, true, true>(s, pr, kf, vf, qf, o, m, l, c, kring, vring, 0, v_addr(0));
  wide_between<0>(s, m, l, o, 1 >= last - 1, rel(0, 1), c);
  WIDE_T(tb2);
  WIDE_ACC(9, tb1, tb2);
  for (int bb = 1; bb < last; ++bb) {
    step_sync(bb);
    WIDE_T(t0);
    wide_half<0, true, true>(s, pr, kf, vf, qf, o, m, l, c, kring, vring, k_addr(bb + 1), 0);
    WIDE_T(t1);
    wide_between<1>(s, m, l, o, false, 0, c);  // query block 1 only meets the diagonal in the wave's last block
    WIDE_T(t2);
    wide_half<1, true, true>(s, pr, kf, vf, qf, o, m, l, c, kring, vring, 0, v_addr(bb));
    WIDE_T(t3);
    wide_between<0>(s, m, l, o, bb + 1 >= last - 1, rel(0, bb + 1), c);
    WIDE_T(t4);
    WIDE_ACC(3, t0, t1);
    WIDE_ACC(4, t1, t2);
    WIDE_ACC(5, t2, t3);
    WIDE_ACC(6, t3, t4);
    WIDE_ACC(7, 0, 1);
  }
  WIDE_T(tb3);
  WIDE_ACC(10, tb2, tb3);
  // the wave's last block: no further scores of query block 0
  step_sync(last);
  wide_half<0, true, true>(s, pr, kf, vf, qf, o, m, l, c, kring, vring, k_addr(last + 1), 0);
  wide_between<1>(s, m, l, o, true, rel(1, last), c);
  wide_half<1, false, true>(s, pr, kf, vf, qf, o, m, l, c, kring, vring, 0, v_addr(last));
  WIDE_T(tb4);
  WIDE_ACC(11, tb3, tb4);
  for (int r = nb_w; r < nb_wg; ++r) step_sync(r);  // the other wave's remaining steps: its blocks still come through this wave's DMA
  __syncthreads();  // nobody reads the rings any more: they become the output strips
  WIDE_T(tb5);
  WIDE_ACC(12, tb4, tb5);
  wide_finish(p, strip, o, pr, vf, m, l, b, h, qw, lane);
  WIDE_T(tb6);
  WIDE_ACC(13, tb5, tb6);
  WIDE_ACC(14, 0, 1);
}

// Registers: everything lives in (at most 256) architectural registers -- the score accumulators are read by the
// vector ALU, so the matrix instructions write ordinary registers.
__global__ __launch_bounds__(kWideThreads, 2) void attn_fwd_wide_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[kWideLdsBytes];
  const int side = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);  // which wave pair
#ifdef WIDE_STAMP
  unsigned long long wide_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  WIDE_T(t_kernel0);
  const unsigned long long rt_kernel0 = wall_clock64();
#endif
  const int nqb = p.T / BQ, per_head = (nqb + 1) >> 1, n_items = per_head * ((p.B * p.H) >> 1);  // an item: one block pair of TWO heads
  for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    int pair, bh2;
    item_of_index(idx, n_items, per_head, pair, bh2);
    const int hi = nqb - 1 - pair;
    for (int half = 0; half < (pair != hi ? 2 : 1); ++half)
      attn_fwd_wide_block(p, smem + side * kWidePairLds, 2 * bh2 + side, half == 0 ? hi : pair WIDE_STAMP_PASS);
  }
#ifdef WIDE_STAMP
  WIDE_T(t_kernel1);
  wide_acc[15] = t_kernel1 - t_kernel0;
  wide_acc[16] = wall_clock64() - rt_kernel0;  // 100 MHz
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) {
    for (int i = 0; i < 20; ++i) g_wide_stamps[blockIdx.x * 4 + (threadIdx.x >> 6)][i] = wide_acc[i];
  }
#endif
}


// =====================================================================================
// Backward.  Two kernels, no atomics, no transposes through memory:
//   attn_bwd_dq      first computes delta[b,h,t] = sum_d dO * O for its queries (and stores it for
//                    the second kernel); same orientation as the forward (query on the lane):
//                    S^T = K Q^T and dP^T = V dO^T share K / V row fragments, lse and delta are
//                    per-lane scalars,
//                    dS^T accumulators are the B operand of dQ^T[d][q] += K^T dS^T, whose A operand
//                    is the transposing read of the same K tile.
//   attn_bwd_dkdv    key on the lane: S = Q K^T and dP = dO V^T with the wave's K / V fragments
//                    resident in registers for the whole kernel; P and dS accumulators are the B
//                    operands of dV^T[d][key] += dO^T P and dK^T[d][key] += Q^T dS, whose A operands
//                    are transposing reads of the dO / Q tiles.  A workgroup owns 128 keys and
//                    walks the query tiles from the diagonal down.
// Both big kernels recompute P from the saved log-sum-exp; together they issue 7 matrix products
// per (query, key) tile pair instead of 5, which buys the absence of a cross-workgroup sum.
// =====================================================================================
struct AttnBwdParams {
  const u16* q;
  const u16* k;
  const u16* v;
  const u16* o;       // [B, T, H, 64]
  const u16* d_o;     // [B, T, H, 64]
  const float* lse;   // [B, H, T]
  float* delta;       // [B, H, T]: sum_d dO*O, written by the dQ kernel and read by the dK/dV kernel
  float* lse2;        // [B, H, T]: lse * log2(e), likewise
  float* bias_partial;  // nullptr, or [B*T/128][3*H*64]: column sums of dq | dk | dv per block of 128 positions
  u16* dq;            // gradient tensors use grad_row_stride between consecutive t
  u16* dk;
  u16* dv;
  long long row_stride, grad_row_stride;
  int B, H, T;
  float scale, scale_log2;
  int per_cu_dq, per_cu_dkdv;  // resident workgroups per CU of the two launches (priority balancing)
};

// Row fragment (A operand: row = tile row 16*t + li, k = 32*kk + 8*g ..) of a [rows][64] tile in the v_off image.
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int t, int kk, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + v_off(16 * t + (lane & 15), 4 * kk + (lane >> 4)));
}

template <bool MASK>
__device__ __forceinline__ void dq_tile(const char* ks, const char* vs, const bf16x8 (&qf)[2][2], const bf16x8 (&dof)[2][2],
                                        f32x4 (&dq)[2][4], const float (&lse2)[2], const float (&dl)[2], int key0, int q0, int lane,
                                        float c) {
  const int g = lane >> 4, li = lane & 15;
  bf16x8 kf[4][2], vf[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[t][kk] = row_frag(ks, t, kk, lane);
      vf[t][kk] = row_frag(vs, t, kk, lane);
    }
  bf16x8 dsf[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    f32x4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][kk], qf[st][kk], s[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t][kk], dof[st][kk], dp[t], 0, 0, 0);
      }
    }
    const int rel = q0 + 16 * st + li - key0 - 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = fast_exp2(fmaf(s[t][e], c, -lse2[st]));
        if (MASK && (16 * t + e > rel)) pe = 0.f;
        s[t][e] = pe * (dp[t][e] - dl[st]);  // the softmax scale is applied once, to the finished dQ rows
      }
#pragma unroll
    for (int sk = 0; sk < 2; ++sk)
      dsf[st][sk] = pack8(s[2 * sk][0], s[2 * sk][1], s[2 * sk][2], s[2 * sk][3], s[2 * sk + 1][0], s[2 * sk + 1][1],
                          s[2 * sk + 1][2], s[2 * sk + 1][3]);
  }
#pragma unroll
  for (int td = 0; td < 4; ++td)
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
      const bf16x8 kt = tr_frag(ks, 32 * sk, td, lane);
#pragma unroll
      for (int st = 0; st < 2; ++st) dq[st][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf[st][sk], dq[st][td], 0, 0, 0);
    }
}

__device__ __forceinline__ void attn_bwd_dq_block(const AttnBwdParams& p, char* smem, int bh, int qb, int round) {
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int q0 = qb * BQ + wid * 32;

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const u16* K = p.k + head_off;
  const u16* V = p.v + head_off;
  const long long o_stride = static_cast<long long>(p.H) * HD;
  const u16* DO = p.d_o + static_cast<long long>(b) * p.T * o_stride + static_cast<long long>(h) * HD;
  const long long stat_off = (static_cast<long long>(b) * p.H + h) * p.T;

  f32x4 dq[2][4];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int td = 0; td < 4; ++td) dq[st][td] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Same three-stage LDS-DMA ring as the forward kernel; both tiles use the v_off image here
  // because K is read row-wise (S^T) and transposed (dQ^T) from the same copy.
  const int n_tiles = (qb * BQ + BQ) / BKV;
  const bool moves_v = wid >= 2;
  const int piece0 = (wid & 1) * 4;
  const u16* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) src[i] = dma_src<true>(moves_v ? V : K, p.row_stride, piece0 + i, lane);
  const long long tile_step = static_cast<long long>(BKV) * p.row_stride;
  const unsigned dst0 = lds_address(smem) + (moves_v ? kTileBytes : 0) + piece0 * 1024;
  auto request = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma_16B(src[i], dst0 + stage * 2 * kTileBytes + i * 1024);
      src[i] += tile_step;
    }
  };
  __syncthreads();  // the previous block's output strips (same memory) have been read back by every wave
  request(0);
  if (n_tiles > 1) request(1);
  // The block's own rows are loaded after the first tiles were requested: one trip to memory before the
  // first product, not two.
  const u16* O = p.o + static_cast<long long>(b) * p.T * o_stride + static_cast<long long>(h) * HD;
  bf16x8 qf[2][2], dof[2][2];
  float lse2[2], dl[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int qi = q0 + 16 * st + li;
    float part = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[st][kk] = *reinterpret_cast<const bf16x8*>(Q + static_cast<long long>(qi) * p.row_stride + 32 * kk + 8 * g);
      dof[st][kk] = *reinterpret_cast<const bf16x8*>(DO + static_cast<long long>(qi) * o_stride + 32 * kk + 8 * g);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(O + static_cast<long long>(qi) * o_stride + 32 * kk + 8 * g);
#pragma unroll
      for (int j = 0; j < 8; ++j) part = fmaf(static_cast<float>(dof[st][kk][j]), static_cast<float>(of[j]), part);
    }
    lse2[st] = p.lse[stat_off + qi] * 1.4426950408889634f;
    // delta = sum_d dO*O of this query: the four lane groups hold 16 d each
    dl[st] = groups_sum(part);
    if (g == 0) {
      p.delta[stat_off + qi] = dl[st];
      p.lse2[stat_off + qi] = lse2[st];
    }
  }
  wait_all_vmem();
  int stage = 0;
  for (int jt = 0; jt < n_tiles; ++jt) {
    take_turns(jt + round);
    if (jt + 1 < n_tiles) wait_vmem_at_most<4>();
    else wait_vmem_at_most<0>();
    __syncthreads();
    if (jt + 2 < n_tiles) request(stage == 0 ? 2 : stage - 1);
    const char* ks = smem + stage * 2 * kTileBytes;
    const char* vs = ks + kTileBytes;
    const int key0 = jt * BKV;
    if (key0 <= q0 + 31) {
      if (key0 + BKV - 1 > q0) dq_tile<true>(ks, vs, qf, dof, dq, lse2, dl, key0, q0, lane, p.scale_log2);
      else dq_tile<false>(ks, vs, qf, dof, dq, lse2, dl, key0, q0, lane, p.scale_log2);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  __syncthreads();  // tile stages -> output strips
  const float by_scale[2] = {p.scale, p.scale};
  const long long C3 = 3LL * p.H * HD;
  float* red = reinterpret_cast<float*>(smem + 4 * kOutStripBytes);  // behind the four strips
  store_rows_staged<true>(smem + wid * kOutStripBytes, dq, by_scale,
                    p.dq + (static_cast<long long>(b) * p.T + q0) * p.grad_row_stride + static_cast<long long>(h) * HD, p.grad_row_stride, lane,
                    p.bias_partial ? red + wid * 64 : nullptr);
  if (p.bias_partial) {
    float* const dst[2] = {p.bias_partial + ((static_cast<long long>(b) * p.T + qb * BQ) >> 7) * C3 + static_cast<long long>(h) * HD, nullptr};
    emit_block_colsums(red, 1, dst, lane, wid);
  }
}

__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[6 * kTileBytes];  // [3 stages][K | V]
  const int round = dispatch_round(p.per_cu_dq);
  const int nqb = p.T / BQ, per_head = (nqb + 1) >> 1, n_items = per_head * p.B * p.H;
  for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    int pair, bh;
    item_of_index(idx, n_items, per_head, pair, bh);
    const int hi = nqb - 1 - pair;
    for (int half = 0; half < (pair != hi ? 2 : 1); ++half) attn_bwd_dq_block(p, smem, bh, half == 0 ? hi : pair, round);
  }
}

constexpr int BQT = 64;                          // queries per LDS stage in the dK/dV kernel (two 32-query halves)
constexpr int kQTileBytes = BQT * HD * 2;        // 8 KiB
constexpr int kQStageBytes = 2 * kQTileBytes + 2 * BQT * 4;  // Q | dO | lse*log2(e) | delta

// 64 consecutive floats -> 256 bytes of LDS, one per lane.
__device__ __forceinline__ void dma_4B(const float* src, unsigned lds_addr) {
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(a) : "memory");
}

template <bool MASK>
__device__ __forceinline__ void dkdv_tile(const char* qs_, const char* dos_, const bf16x8 (&kfB)[2][2], const bf16x8 (&vfB)[2][2],
                                          f32x4 (&dk)[2][4], f32x4 (&dv)[2][4], const char* stats, int hf, int qt0,
                                          int k0w, int lane, float c) {
  // Works on the 32 queries qt0 .. qt0+31, which are rows 32*hf .. 32*hf+31 of the stage's tiles.
  const int g = lane >> 4, li = lane & 15;
  // per-register query statistics: query = qt0 + 16*qs + 4*g + e
  f32x4 l2[2], dl[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    l2[qs] = *reinterpret_cast<const f32x4*>(stats + (32 * hf + 16 * qs + 4 * g) * 4);
    dl[qs] = *reinterpret_cast<const f32x4*>(stats + BQT * 4 + (32 * hf + 16 * qs + 4 * g) * 4);
  }
  bf16x8 qrow[2][2], dorow[2][2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qrow[qs][kk] = row_frag(qs_, 2 * hf + qs, kk, lane);
      dorow[qs][kk] = row_frag(dos_, 2 * hf + qs, kk, lane);
    }
  bf16x8 pB[2], dsB[2];
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2) {
    f32x4 s[2], dp[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      s[qs] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[qs] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        s[qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qrow[qs][kk], kfB[kt2][kk], s[qs], 0, 0, 0);
        dp[qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dorow[qs][kk], vfB[kt2][kk], dp[qs], 0, 0, 0);
      }
    }
    // lane: key = k0w + 16*kt2 + li ; register (qs, e): query = qt0 + 16*qs + 4*g + e
    const int rel = k0w + 16 * kt2 + li - qt0 - 4 * g;  // masked (future key) when 16*qs + e < rel
    float pv[2][4], dsv[2][4];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = fast_exp2(fmaf(s[qs][e], c, -l2[qs][e]));
        if (MASK && (16 * qs + e < rel)) pe = 0.f;
        pv[qs][e] = pe;
        dsv[qs][e] = pe * (dp[qs][e] - dl[qs][e]);  // the softmax scale is applied once, to the finished dK rows
      }
    pB[kt2] = pack8(pv[0][0], pv[0][1], pv[0][2], pv[0][3], pv[1][0], pv[1][1], pv[1][2], pv[1][3]);
    dsB[kt2] = pack8(dsv[0][0], dsv[0][1], dsv[0][2], dsv[0][3], dsv[1][0], dsv[1][1], dsv[1][2], dsv[1][3]);
  }
  // dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS   (contraction over the tile's 32 queries)
#pragma unroll
  for (int td = 0; td < 4; ++td) {
    const bf16x8 dot = tr_frag(dos_, 32 * hf, td, lane);
    const bf16x8 qt = tr_frag(qs_, 32 * hf, td, lane);
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
      dv[kt2][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pB[kt2], dv[kt2][td], 0, 0, 0);
      dk[kt2][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsB[kt2], dk[kt2][td], 0, 0, 0);
    }
  }
}

// One block of 128 keys of one (batch, head); key block 0 has the most query tiles.
__device__ __forceinline__ void attn_bwd_dkdv_block(const AttnBwdParams& p, char* smem, int bh, int jb, int round) {
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int k0w = jb * BQ + wid * 32;  // this wave's 32 keys

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const u16* K = p.k + head_off;
  const u16* V = p.v + head_off;
  const long long o_stride = static_cast<long long>(p.H) * HD;
  const u16* DO = p.d_o + static_cast<long long>(b) * p.T * o_stride + static_cast<long long>(h) * HD;
  const long long stat_off = (static_cast<long long>(b) * p.H + h) * p.T;
  const float* lse2 = p.lse2 + stat_off;
  const float* delta = p.delta + stat_off;

  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int td = 0; td < 4; ++td) {
      dk[kt2][td] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[kt2][td] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  // Q / dO tiles (64 queries) and the queries' statistics arrive by LDS-DMA through a three-stage
  // ring, as in the other kernels: waves 0-1 move Q pieces, waves 2-3 dO pieces, wave 0 also the 64
  // log-sum-exp values and wave 2 the 64 delta values (one 4-byte DMA each).
  const int first_qt = (jb * BQ) / BQT, n_qt = p.T / BQT;
  const bool moves_do = wid >= 2, moves_stat = (wid & 1) == 0;
  const int piece0 = (wid & 1) * 4;
  const long long src_stride = moves_do ? o_stride : p.row_stride;
  const u16* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    src[i] = dma_src<true>((moves_do ? DO : Q) + static_cast<long long>(first_qt) * BQT * src_stride, src_stride, piece0 + i, lane);
  const float* stat_src = (moves_do ? delta : lse2) + first_qt * BQT + lane;
  const long long tile_step = static_cast<long long>(BQT) * src_stride;
  const unsigned smem_addr = lds_address(smem);
  const unsigned dst0 = smem_addr + (moves_do ? kQTileBytes : 0) + piece0 * 1024;
  const unsigned stat_dst0 = smem_addr + 2 * kQTileBytes + (moves_do ? BQT * 4 : 0);
  auto request = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma_16B(src[i], dst0 + stage * kQStageBytes + i * 1024);
      src[i] += tile_step;
    }
    if (moves_stat) {
      dma_4B(stat_src, stat_dst0 + stage * kQStageBytes);
      stat_src += BQT;
    }
  };
  const int n_iter = n_qt - first_qt;
  __syncthreads();  // the previous block's output strips (same memory) have been read back by every wave
  request(0);
  if (n_iter > 1) request(1);
  // (loaded after the first tiles were requested: one trip to memory before the first product, not two)
  // K / V fragments as B operands (col = key li of key tile kt2, k = d), resident for the whole kernel
  bf16x8 kfB[2][2], vfB[2][2];
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const long long row = k0w + 16 * kt2 + li;
      kfB[kt2][kk] = *reinterpret_cast<const bf16x8*>(K + row * p.row_stride + 32 * kk + 8 * g);
      vfB[kt2][kk] = *reinterpret_cast<const bf16x8*>(V + row * p.row_stride + 32 * kk + 8 * g);
    }
  wait_all_vmem();
  int stage = 0;
  for (int it = 0; it < n_iter; ++it) {
    take_turns(it + round);
    if (it + 1 < n_iter) {  // leave the next stage's requests in flight: 5 per wave with statistics, else 4
      if (moves_stat) wait_vmem_at_most<5>();
      else wait_vmem_at_most<4>();
    } else {
      wait_vmem_at_most<0>();
    }
    __syncthreads();
    if (it + 2 < n_iter) request(stage == 0 ? 2 : stage - 1);
    const char* qs_ = smem + stage * kQStageBytes;
    const char* dos_ = qs_ + kQTileBytes;
    const char* stats = qs_ + 2 * kQTileBytes;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int qt0 = (first_qt + it) * BQT + 32 * hf;
      if (qt0 + 31 >= k0w) {  // otherwise every query of the half precedes all of this wave's keys
        if (qt0 < k0w + 31) dkdv_tile<true>(qs_, dos_, kfB, vfB, dk, dv, stats, hf, qt0, k0w, lane, p.scale_log2);
        else dkdv_tile<false>(qs_, dos_, kfB, vfB, dk, dv, stats, hf, qt0, k0w, lane, p.scale_log2);
      }
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  // lane: key = k0w + 16*kt2 + li, d = 16*td + 4*g + e
  __syncthreads();  // tile stages -> output strips
  const float one[2] = {1.f, 1.f};
  const long long out_off = (static_cast<long long>(b) * p.T + k0w) * p.grad_row_stride + static_cast<long long>(h) * HD;
  const long long C3 = 3LL * p.H * HD;
  float* red = reinterpret_cast<float*>(smem + 4 * kOutStripBytes);  // behind the four strips: one set for dK, one for dV
  const float by_scale[2] = {p.scale, p.scale};
  store_rows_staged<true>(smem + wid * kOutStripBytes, dk, by_scale, p.dk + out_off, p.grad_row_stride, lane,
                          p.bias_partial ? red + wid * 64 : nullptr);
  store_rows_staged<true>(smem + wid * kOutStripBytes, dv, one, p.dv + out_off, p.grad_row_stride, lane,
                          p.bias_partial ? red + kColsumSetFloats + wid * 64 : nullptr);
  if (p.bias_partial) {
    float* bk = p.bias_partial + ((static_cast<long long>(b) * p.T + jb * BQ) >> 7) * C3 + (p.H + h) * static_cast<long long>(HD);
    float* const dst[2] = {bk, bk + static_cast<long long>(p.H) * HD};
    emit_block_colsums(red, 2, dst, lane, wid);
  }
}

__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dkdv_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[3 * kQStageBytes];  // [3 stages][Q | dO | lse2 | delta]
  const int round = dispatch_round(p.per_cu_dkdv);
  const int nkb = p.T / BQ, per_head = (nkb + 1) >> 1, n_items = per_head * p.B * p.H;
  for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    int pair, bh;
    item_of_index(idx, n_items, per_head, pair, bh);
    const int lo = nkb - 1 - pair;
    // key block `pair` sees the most queries, its partner the fewest
    for (int half = 0; half < (pair != lo ? 2 : 1); ++half) attn_bwd_dkdv_block(p, smem, bh, half == 0 ? pair : lo, round);
  }
}

// =====================================================================================
// Backward on v_mfma_f32_32x32x16_bf16.  Same decomposition, rings, work items and outputs as the two kernels above;
// what changes is the matrix instruction: a wave's 32 queries (dQ kernel) or 32 keys (dK/dV kernel) sit on the 32
// accumulator columns, so every product of a 32 x 32 block takes 4 instructions instead of 8.  A matrix instruction
// holds the issue port it shares with the vector ALU for ~20 cycles whatever its shape (tools/kernel_harness/
// mfma_chain_probe.hip: the FLOPs of 8 of these next to 48 vector instructions take 460 cycles, as 16 of the 16x16x32
// kind 592).  Built to test whether that port bounds the backward: it does not (see backward_mode() below for the
// numbers), so these kernels are selectable (kernel = "wide", DPA_ATTN_BWD=wide) but not the default.
// Tiles that are read both row-wise and transposed use one image (x_off): chunk ^ rot3((row >> 1) & 7) -- 16 consecutive
// rows of one chunk cover all banks (row-wise 16-byte reads of a quarter wave), and rows r, r + 2 of a 4-row x 64-byte
// block fall into different 64-byte halves (transposing reads of a half wave).
// =====================================================================================
__device__ __forceinline__ int x_swz(int row) {
  const int p = (row >> 1) & 7;
  return ((p & 1) << 2) | (p >> 1);
}
__device__ __forceinline__ int x_off(int row, int c) { return row * 128 + ((c ^ x_swz(row)) << 4); }
__device__ __forceinline__ const u16* dma_src_x(const u16* base, long long row_stride, int piece, int lane) {
  const int r = piece * 8 + (lane >> 3), pc = lane & 7;
  return base + static_cast<long long>(r) * row_stride + (pc ^ x_swz(r)) * 8;
}
// Row fragment (A or B operand): row = 32*rblk + (lane & 31), k = 16*ds + 8*(lane >> 5) ..
__device__ __forceinline__ bf16x8 row32_frag(const char* tile, int rblk, int ds, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + x_off(32 * rblk + (lane & 31), 2 * ds + (lane >> 5)));
}
// Transposed fragment (A operand X^T): row of X^T = column 32*db + (lane & 31) of the tile; the eight k-slots of lane
// half hi are tile rows row_base + 16*s2 + 4*hi + (0..3) and + 8 + (0..3): the rows a 32x32 accumulator keeps in
// registers 8*s2 .. 8*s2+7 of that half.
__device__ __forceinline__ bf16x8 tr32_frag(const char* tile, int row_base, int s2, int db, int lane) {
  const int i = lane & 15, g1 = (lane >> 4) & 1, hi = lane >> 5, pp = i & 3;
  const int c = 2 * (2 * db + g1) + (pp >> 1), sub = (pp & 1) << 3;
  const int r1 = row_base + 16 * s2 + 4 * hi + (i >> 2), r2 = r1 + 8;
  using lds_ptr = __attribute__((address_space(3))) s16x4*;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + x_off(r1, c) + sub));
  const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + x_off(r2, c) + sub));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8_of(const f32x16& a, int s2) {
  return pack8(a[8 * s2], a[8 * s2 + 1], a[8 * s2 + 2], a[8 * s2 + 3], a[8 * s2 + 4], a[8 * s2 + 5], a[8 * s2 + 6], a[8 * s2 + 7]);
}
// 32 output rows x 64 d of a wave (acc[db][i]: row = lane & 31, d = 32*db + (i & 3) + 8*(i >> 2) + 4*(lane >> 5)) through
// the wave's LDS strip, as store_rows_staged does for the 16x16 layout; `colsum` likewise.
template <bool STREAM>
__device__ __forceinline__ void store_rows32_staged(char* strip, const f32x16 (&acc)[2], float row_scale, u16* out, long long row_stride,
                                                    int lane, float* colsum = nullptr) {
  const int hi = lane >> 5, lq = lane & 31;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = BF16::from_f32(acc[db][4 * j + e] * row_scale);
      *reinterpret_cast<u16x4*>(strip + lq * kOutPitch + (32 * db + 8 * j + 4 * hi) * 2) = w;
    }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = 8 * it + (lane >> 3), ch = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(strip + r * kOutPitch + ch * 16);
    if (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + r * row_stride + ch * 8));
    else *reinterpret_cast<u32x4*>(out + r * row_stride + ch * 8) = v;
  }
  if (colsum) {
    float cs = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) cs += BF16::to_f32(*reinterpret_cast<const u16*>(strip + r * kOutPitch + lane * 2));
    colsum[lane] = cs;
  }
}

// dQ: one 64-key tile for the wave's 32 queries (lane & 31 = query).
template <bool MASK>
__device__ __forceinline__ void dq32_tile(const char* ks, const char* vs, const bf16x8 (&qf)[4], const bf16x8 (&dof)[4], f32x16 (&dq)[2],
                                          float lse2, float dl, int key0, int q0, int lane, float c) {
  const int hi = lane >> 5, lq = lane & 31;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    if (MASK && key0 + 32 * kb > q0 + 31) break;  // every key of the block is in the future of every query of the wave
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = 0.f;
      dp[e] = 0.f;
    }
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row32_frag(ks, kb, ds, lane), qf[ds], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row32_frag(vs, kb, ds, lane), dof[ds], dp, 0, 0, 0);
    }
    const int rel = q0 + lq - key0 - 32 * kb - 4 * hi;  // register i holds key (i & 3) + 8 * (i >> 2) of the lane half
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float pe = fast_exp2(fmaf(s[i], c, -lse2));
      if (MASK && ((i & 3) + 8 * (i >> 2) > rel)) pe = 0.f;
      s[i] = pe * (dp[i] - dl);  // the softmax scale is applied once, to the finished dQ rows
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 dsf = pack8_of(s, s2);
#pragma unroll
      for (int db = 0; db < 2; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr32_frag(ks, 32 * kb, s2, db, lane), dsf, dq[db], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void attn_bwd_dq32_block(const AttnBwdParams& p, char* smem, int bh, int qb, int round) {
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  const int q0 = qb * BQ + wid * 32;

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const u16* K = p.k + head_off;
  const u16* V = p.v + head_off;
  const long long o_stride = static_cast<long long>(p.H) * HD;
  const u16* DO = p.d_o + static_cast<long long>(b) * p.T * o_stride + static_cast<long long>(h) * HD;
  const u16* O = p.o + static_cast<long long>(b) * p.T * o_stride + static_cast<long long>(h) * HD;
  const long long stat_off = (static_cast<long long>(b) * p.H + h) * p.T;

  f32x16 dq[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[db][e] = 0.f;

  // the same three-stage LDS-DMA ring as in attn_bwd_dq_block; both tiles in the x_off image
  const int n_tiles = (qb * BQ + BQ) / BKV;
  const bool moves_v = wid >= 2;
  const int piece0 = (wid & 1) * 4;
  const u16* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) src[i] = dma_src_x(moves_v ? V : K, p.row_stride, piece0 + i, lane);
  const long long tile_step = static_cast<long long>(BKV) * p.row_stride;
  const unsigned dst0 = lds_address(smem) + (moves_v ? kTileBytes : 0) + piece0 * 1024;
  auto request = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma_16B(src[i], dst0 + stage * 2 * kTileBytes + i * 1024);
      src[i] += tile_step;
    }
  };
  __syncthreads();  // the previous block's output strips (same memory) have been read back by every wave
  request(0);
  if (n_tiles > 1) request(1);
  // the wave's own rows (B operands: lane & 31 = query, d = 16*ds + 8*hi ..) and delta = sum_d dO * O
  bf16x8 qf[4], dof[4];
  const int qi = q0 + lq;
  float part = 0.f;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    qf[ds] = *reinterpret_cast<const bf16x8*>(Q + static_cast<long long>(qi) * p.row_stride + 16 * ds + 8 * hi);
    dof[ds] = *reinterpret_cast<const bf16x8*>(DO + static_cast<long long>(qi) * o_stride + 16 * ds + 8 * hi);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(O + static_cast<long long>(qi) * o_stride + 16 * ds + 8 * hi);
#pragma unroll
    for (int j = 0; j < 8; ++j) part = fmaf(static_cast<float>(dof[ds][j]), static_cast<float>(of[j]), part);
  }
  const float lse2 = p.lse[stat_off + qi] * 1.4426950408889634f;
  const float dl = half_swap_sum(part);  // the two lane halves hold 32 d each
  if (hi == 0) {
    p.delta[stat_off + qi] = dl;
    p.lse2[stat_off + qi] = lse2;
  }
  wait_all_vmem();
  int stage = 0;
  for (int jt = 0; jt < n_tiles; ++jt) {
    take_turns(jt + round);
    if (jt + 1 < n_tiles) wait_vmem_at_most<4>();
    else wait_vmem_at_most<0>();
    __syncthreads();
    if (jt + 2 < n_tiles) request(stage == 0 ? 2 : stage - 1);
    const char* ks = smem + stage * 2 * kTileBytes;
    const char* vs = ks + kTileBytes;
    const int key0 = jt * BKV;
    if (key0 <= q0 + 31) {
      if (key0 + BKV - 1 > q0) dq32_tile<true>(ks, vs, qf, dof, dq, lse2, dl, key0, q0, lane, p.scale_log2);
      else dq32_tile<false>(ks, vs, qf, dof, dq, lse2, dl, key0, q0, lane, p.scale_log2);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  __syncthreads();  // tile stages -> output strips
  const long long C3 = 3LL * p.H * HD;
  float* red = reinterpret_cast<float*>(smem + 4 * kOutStripBytes);
  store_rows32_staged<true>(smem + wid * kOutStripBytes, dq, p.scale,
                            p.dq + (static_cast<long long>(b) * p.T + q0) * p.grad_row_stride + static_cast<long long>(h) * HD,
                            p.grad_row_stride, lane, p.bias_partial ? red + wid * 64 : nullptr);
  if (p.bias_partial) {
    float* const dst[2] = {p.bias_partial + ((static_cast<long long>(b) * p.T + qb * BQ) >> 7) * C3 + static_cast<long long>(h) * HD, nullptr};
    emit_block_colsums(red, 1, dst, lane, wid);
  }
}

__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dq32_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[6 * kTileBytes];  // [3 stages][K | V]
  const int round = dispatch_round(p.per_cu_dq);
  const int nqb = p.T / BQ, per_head = (nqb + 1) >> 1, n_items = per_head * p.B * p.H;
  for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    int pair, bh;
    item_of_index(idx, n_items, per_head, pair, bh);
    const int hi = nqb - 1 - pair;
    for (int half = 0; half < (pair != hi ? 2 : 1); ++half) attn_bwd_dq32_block(p, smem, bh, half == 0 ? hi : pair, round);
  }
}

// dK / dV: the 32 queries qt0 .. qt0+31 (rows 32*hf .. of the stage's tiles) for the wave's 32 keys (lane & 31 = key).
template <bool MASK>
__device__ __forceinline__ void dkdv32_tile(const char* qs_, const char* dos_, const bf16x8 (&kfB)[4], const bf16x8 (&vfB)[4], f32x16 (&dk)[2],
                                            f32x16 (&dv)[2], const char* stats, int hf, int qt0, int k0w, int lane, float c) {
  const int hi = lane >> 5, lk = lane & 31;
  f32x16 s, dp;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    s[e] = 0.f;
    dp[e] = 0.f;
  }
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row32_frag(qs_, hf, ds, lane), kfB[ds], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row32_frag(dos_, hf, ds, lane), vfB[ds], dp, 0, 0, 0);
  }
  // register i: query = qt0 + (i & 3) + 8 * (i >> 2) + 4 * hi; its statistics come four queries at a time
  const int rel = k0w + lk - qt0 - 4 * hi;  // masked (the key is in the query's future) when (i & 3) + 8 * (i >> 2) < rel
  f32x16 pv;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 l2 = *reinterpret_cast<const f32x4*>(stats + (32 * hf + 8 * j + 4 * hi) * 4);
    const f32x4 dl = *reinterpret_cast<const f32x4*>(stats + BQT * 4 + (32 * hf + 8 * j + 4 * hi) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = fast_exp2(fmaf(s[4 * j + e], c, -l2[e]));
      if (MASK && (8 * j + e < rel)) pe = 0.f;
      pv[4 * j + e] = pe;
      s[4 * j + e] = pe * (dp[4 * j + e] - dl[e]);  // the softmax scale is applied once, to the finished dK rows
    }
  }
  // dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS   (contraction over the 32 queries)
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const bf16x8 pB = pack8_of(pv, s2), dsB = pack8_of(s, s2);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr32_frag(dos_, 32 * hf, s2, db, lane), pB, dv[db], 0, 0, 0);
      dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr32_frag(qs_, 32 * hf, s2, db, lane), dsB, dk[db], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void attn_bwd_dkdv32_block(const AttnBwdParams& p, char* smem, int bh, int jb, int round) {
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int hi = lane >> 5, lk = lane & 31;
  const int k0w = jb * BQ + wid * 32;  // this wave's 32 keys

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const u16* K = p.k + head_off;
  const u16* V = p.v + head_off;
  const long long o_stride = static_cast<long long>(p.H) * HD;
  const u16* DO = p.d_o + static_cast<long long>(b) * p.T * o_stride + static_cast<long long>(h) * HD;
  const long long stat_off = (static_cast<long long>(b) * p.H + h) * p.T;
  const float* lse2 = p.lse2 + stat_off;
  const float* delta = p.delta + stat_off;

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dk[db][e] = 0.f;
      dv[db][e] = 0.f;
    }

  // the same three-stage ring of Q / dO tiles (64 queries) and statistics as in attn_bwd_dkdv_block; tiles in the x_off image
  const int first_qt = (jb * BQ) / BQT, n_qt = p.T / BQT;
  const bool moves_do = wid >= 2, moves_stat = (wid & 1) == 0;
  const int piece0 = (wid & 1) * 4;
  const long long src_stride = moves_do ? o_stride : p.row_stride;
  const u16* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    src[i] = dma_src_x((moves_do ? DO : Q) + static_cast<long long>(first_qt) * BQT * src_stride, src_stride, piece0 + i, lane);
  const float* stat_src = (moves_do ? delta : lse2) + first_qt * BQT + lane;
  const long long tile_step = static_cast<long long>(BQT) * src_stride;
  const unsigned smem_addr = lds_address(smem);
  const unsigned dst0 = smem_addr + (moves_do ? kQTileBytes : 0) + piece0 * 1024;
  const unsigned stat_dst0 = smem_addr + 2 * kQTileBytes + (moves_do ? BQT * 4 : 0);
  auto request = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma_16B(src[i], dst0 + stage * kQStageBytes + i * 1024);
      src[i] += tile_step;
    }
    if (moves_stat) {
      dma_4B(stat_src, stat_dst0 + stage * kQStageBytes);
      stat_src += BQT;
    }
  };
  const int n_iter = n_qt - first_qt;
  __syncthreads();  // the previous block's output strips (same memory) have been read back by every wave
  request(0);
  if (n_iter > 1) request(1);
  // K / V fragments as B operands (lane & 31 = key, d = 16*ds + 8*hi ..), resident for the whole block
  bf16x8 kfB[4], vfB[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    const long long row = k0w + lk;
    kfB[ds] = *reinterpret_cast<const bf16x8*>(K + row * p.row_stride + 16 * ds + 8 * hi);
    vfB[ds] = *reinterpret_cast<const bf16x8*>(V + row * p.row_stride + 16 * ds + 8 * hi);
  }
  wait_all_vmem();
  int stage = 0;
  for (int it = 0; it < n_iter; ++it) {
    take_turns(it + round);
    if (it + 1 < n_iter) {  // leave the next stage's requests in flight: 5 per wave with statistics, else 4
      if (moves_stat) wait_vmem_at_most<5>();
      else wait_vmem_at_most<4>();
    } else {
      wait_vmem_at_most<0>();
    }
    __syncthreads();
    if (it + 2 < n_iter) request(stage == 0 ? 2 : stage - 1);
    const char* qs_ = smem + stage * kQStageBytes;
    const char* dos_ = qs_ + kQTileBytes;
    const char* stats = qs_ + 2 * kQTileBytes;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int qt0 = (first_qt + it) * BQT + 32 * hf;
      if (qt0 + 31 >= k0w) {  // otherwise every query of the half precedes all of this wave's keys
        if (qt0 < k0w + 31) dkdv32_tile<true>(qs_, dos_, kfB, vfB, dk, dv, stats, hf, qt0, k0w, lane, p.scale_log2);
        else dkdv32_tile<false>(qs_, dos_, kfB, vfB, dk, dv, stats, hf, qt0, k0w, lane, p.scale_log2);
      }
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  __syncthreads();  // tile stages -> output strips
  const long long out_off = (static_cast<long long>(b) * p.T + k0w) * p.grad_row_stride + static_cast<long long>(h) * HD;
  const long long C3 = 3LL * p.H * HD;
  float* red = reinterpret_cast<float*>(smem + 4 * kOutStripBytes);
  store_rows32_staged<true>(smem + wid * kOutStripBytes, dk, p.scale, p.dk + out_off, p.grad_row_stride, lane,
                            p.bias_partial ? red + wid * 64 : nullptr);
  store_rows32_staged<true>(smem + wid * kOutStripBytes, dv, 1.f, p.dv + out_off, p.grad_row_stride, lane,
                            p.bias_partial ? red + kColsumSetFloats + wid * 64 : nullptr);
  if (p.bias_partial) {
    float* bk = p.bias_partial + ((static_cast<long long>(b) * p.T + jb * BQ) >> 7) * C3 + (p.H + h) * static_cast<long long>(HD);
    float* const dst[2] = {bk, bk + static_cast<long long>(p.H) * HD};
    emit_block_colsums(red, 2, dst, lane, wid);
  }
}

__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dkdv32_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[3 * kQStageBytes];  // [3 stages][Q | dO | lse2 | delta]
  const int round = dispatch_round(p.per_cu_dkdv);
  const int nkb = p.T / BQ, per_head = (nkb + 1) >> 1, n_items = per_head * p.B * p.H;
  for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    int pair, bh;
    item_of_index(idx, n_items, per_head, pair, bh);
    const int lo = nkb - 1 - pair;
    for (int half = 0; half < (pair != lo ? 2 : 1); ++half) attn_bwd_dkdv32_block(p, smem, bh, half == 0 ? pair : lo, round);
  }
}

// =====================================================================================
// Forward on v_mfma_f32_32x32x16_bf16 with the FIRST kernel's structure (32 queries per wave, 64-key tiles through the
// same three-stage LDS-DMA ring, one barrier per tile, all overlap left to the waves that share a SIMD): 16 matrix
// instructions per 32 x 64 tile instead of 36.  A matrix instruction holds the issue port it shares with the vector ALU
// for ~20 cycles whatever its shape, and the first kernel's tile loop runs within 10 % of what that port allows (36 x 20
// + ~640 cycles of vector work per wave and tile against 1465 measured), so this is the direct remedy; the wave also
// needs fewer registers; a third workgroup per CU (168 registers, a few spills) measured the same as two.
// Measured: 210-216 us at B64 x H12 x T1024 against 213-220 us for the first kernel and 207-210 us for the 64-query one;
// 30.3 vs 31.9 us at B8; 91-94 vs 95 us at B96 x T512 -- a few per cent, not the 29 % the port arithmetic promised:
// with matrix, vector and transcendental instructions sharing a SIMD's issue, all three forward kernels end up within
// 3 % of each other.
// K tile in the k_off image (row-wise reads), V tile in the v2_off image (transposing reads), as in the 64-query kernel.
// =====================================================================================
__device__ __forceinline__ const u16* dma_src_v2(const u16* base, long long row_stride, int piece, int lane) {
  const int r = piece * 8 + (lane >> 3), pc = lane & 7;
  return base + static_cast<long long>(r) * row_stride + (pc ^ (((r >> 1) & 1) << 2)) * 8;
}

template <bool MASK>
__device__ __forceinline__ void fwd32_tile(const char* ks, const char* vs, const bf16x8 (&qf)[4], f32x16 (&o)[2], float& m, float& l,
                                           int key0, int q0, int lane, float c, int koff, int voff) {
  const int hi = lane >> 5, lq = lane & 31;
  bf16x2 ones;
  ones[0] = static_cast<__bf16>(1.f);
  ones[1] = static_cast<__bf16>(1.f);
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    if (MASK && key0 + 32 * kb > q0 + 31) break;  // every key of the block is in the future of every query of the wave
    bf16x8 kf[4];
    wide_load_k(ks, kb * kBlkBytes + koff, kf);
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds], qf[ds], s, 0, 0, 0);
    bf16x8 vf[2][2];
    wide_load_v(vs, kb * kBlkBytes + voff, vf);
    if (MASK) {
      const int rel = q0 + lq - key0 - 32 * kb - 4 * hi;  // register i holds key (i & 3) + 8 * (i >> 2) of the lane half
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if ((i & 3) + 8 * (i >> 2) > rel) s[i] = -__builtin_huge_valf();
    }
    float t = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) t = fmaxf(fmaxf(t, s[i]), s[i + 1]);
    const float mx = fmaxf(t, s[15]) * c;
    if (__any(mx > m + kDeferLog2)) {  // rare after the first tile: some query's maximum moved a lot
      const float mg = half_swap_max(mx);
      const float m_new = mg > m + kDeferLog2 ? mg : m;
      const float alpha = m_new == m ? 1.f : fast_exp2(m - m_new);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
    }
    const float nm = -m;
    bf16x8 pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      unsigned w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x2 v;
        v[0] = static_cast<__bf16>(fast_exp2(fmaf(s[8 * s2 + 2 * j], c, nm)));
        v[1] = static_cast<__bf16>(fast_exp2(fmaf(s[8 * s2 + 2 * j + 1], c, nm)));
        l = __builtin_amdgcn_fdot2_f32_bf16(v, ones, l, false);
        w[j] = __builtin_bit_cast(unsigned, v);
      }
      const u32x4 packed = {w[0], w[1], w[2], w[3]};
      pf[s2] = __builtin_bit_cast(bf16x8, packed);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][s2], pf[s2], o[db], 0, 0, 0);
  }
}

__device__ __forceinline__ void attn_fwd32_block(const AttnParams& p, char* smem, int bh, int qb, int round) {
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  const int q0 = qb * BQ + wid * 32;

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const u16* K = p.k + head_off;
  const u16* V = p.v + head_off;

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = -__builtin_huge_valf(), l = 0.f;

  const int koff = k_off(lq, hi);
  int voff;
  {
    const int i = lane & 15, g1 = (lane >> 4) & 1, pp = i & 3, row = 4 * hi + (i >> 2);
    voff = v2_off(row, 2 * g1 + (pp >> 1)) + ((pp & 1) << 3);
  }
  // the first kernel's ring: waves 0-1 move K pieces, waves 2-3 V pieces, four 1-KiB pieces per wave and tile
  const int n_tiles = (qb * BQ + BQ) / BKV;
  const bool moves_v = wid >= 2;
  const int piece0 = (wid & 1) * 4;
  const u16* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    src[i] = moves_v ? dma_src_v2(V, p.row_stride, piece0 + i, lane) : dma_src<false>(K, p.row_stride, piece0 + i, lane);
  const long long tile_step = static_cast<long long>(BKV) * p.row_stride;
  const int dst0 = (moves_v ? kTileBytes : 0) + piece0 * 1024;
  const unsigned smem_addr = lds_address(smem);
  auto request = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma_16B(src[i], smem_addr + stage * 2 * kTileBytes + dst0 + i * 1024);
      src[i] += tile_step;
    }
  };
  __syncthreads();  // the previous block's output strips (same memory) have been read back by every wave
  request(0);
  if (n_tiles > 1) request(1);
  bf16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(Q + static_cast<long long>(q0 + lq) * p.row_stride + 16 * ds + 8 * hi);
  wait_all_vmem();

  int stage = 0;
  for (int jt = 0; jt < n_tiles; ++jt) {
    take_turns(jt + round);
    if (jt + 1 < n_tiles) wait_vmem_at_most<4>();
    else wait_vmem_at_most<0>();
    __syncthreads();
    if (jt + 2 < n_tiles) request(stage == 0 ? 2 : stage - 1);
    const char* ks = smem + stage * 2 * kTileBytes;
    const char* vs = ks + kTileBytes;
    const int key0 = jt * BKV;
    if (key0 <= q0 + 31) {
      if (key0 + BKV - 1 > q0) fwd32_tile<true>(ks, vs, qf, o, m, l, key0, q0, lane, p.scale_log2, koff, voff);
      else fwd32_tile<false>(ks, vs, qf, o, m, l, key0, q0, lane, p.scale_log2, koff, voff);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  const float lt = half_swap_sum(l);
  if (hi == 0) p.lse[(static_cast<long long>(b) * p.H + h) * p.T + q0 + lq] = (m + log2f(lt)) * 0.6931471805599453f;
  __syncthreads();  // all waves are done with the tile stages: their memory becomes the output strips
  const long long o_stride = static_cast<long long>(p.H) * HD;
  store_rows32_staged<true>(smem + wid * kOutStripBytes, o, 1.f / lt, p.o + (static_cast<long long>(b) * p.T + q0) * o_stride + static_cast<long long>(h) * HD,
                            o_stride, lane);
}

__global__ __launch_bounds__(kThreads, FWD32_WAVES_PER_SIMD) void attn_fwd32_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[6 * kTileBytes];  // [3 stages][K | V]
  const int round = dispatch_round(p.per_cu);
  const int nqb = p.T / BQ, per_head = (nqb + 1) >> 1, n_items = per_head * p.B * p.H;
  for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
    int pair, bh;
    item_of_index(idx, n_items, per_head, pair, bh);
    const int hi = nqb - 1 - pair;
    for (int half = 0; half < (pair != hi ? 2 : 1); ++half) attn_fwd32_block(p, smem, bh, half == 0 ? hi : pair, round);
  }
}

}  // namespace

namespace {
// Workgroups of a persistent launch: exactly as many as are resident at once for this kernel (asked of
// the runtime, so register or LDS growth cannot silently turn one round into one and a bit), capped by
// the number of work items, and a multiple of 8 so that the items a workgroup strides over all belong
// to its own XCD's run.
template <typename Kernel>
int resident_per_cu(Kernel kernel, int threads = kThreads) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  return per_cu;
}
// DPA_ATTN_FWD=classic|wide|tile32 forces one forward kernel (default: wide where it pays, else classic -- see launch_attention_fwd);
// DPA_ATTN_WIDE_PER_CU caps the wide kernel's workgroups per CU.
int wide_forward_mode() {
  const char* e = getenv("DPA_ATTN_FWD");
  if (e && e[0] == 'c') return 0;
  if (e && e[0] == 'w') return 2;
  if (e && e[0] == 't') return 3;  // "tile32"
  return 1;
}
// DPA_ATTN_BWD=classic|wide forces the backward kernels on 16x16x32 / 32x32x16 matrix instructions.  Default: the former --
// measured on the same boxes the 32x32x16 pair is 3-6 % slower alone (545 vs 515 us at B64 H12 T1024) and 0.3 % slower
// inside the GPT-2 step, although it issues half as many matrix instructions: these kernels wait on their dependent
// chains (fragment reads -> products -> exponentials -> products), not on the issue port.
int backward_mode() {
  const char* e = getenv("DPA_ATTN_BWD");
  return (e && e[0] == 'w') ? 1 : 0;
}
int wide_per_cu() {
  int n = resident_per_cu(attn_fwd_wide_kernel, kWideThreads);
  const char* e = getenv("DPA_ATTN_WIDE_PER_CU");
  if (e && atoi(e) > 0 && atoi(e) < n) n = atoi(e);
  return n;
}
int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return cus;
}
unsigned persistent_grid(int per_cu, int T, int B, int H, int heads_per_item = 1) {
  const long long items = static_cast<long long>((T / BQ + 1) / 2) * B * H / heads_per_item;
  long long g = static_cast<long long>(cu_count()) * per_cu;
  if (g > items) g = items;
  g = g >= 8 ? g / 8 * 8 : g;
  return static_cast<unsigned>(g < 1 ? 1 : g);
}
}  // namespace

bool attention_supported(long long T, long long head_dim, long long row_stride) {
  return head_dim == HD && T % BQ == 0 && T >= BQ && row_stride % 8 == 0;
}

void launch_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, long long row_stride, int B, int H,
                          int T, float softmax_scale, hipStream_t stream, int kernel) {
  AttnParams p;
  p.q = static_cast<const u16*>(q);
  p.k = static_cast<const u16*>(k);
  p.v = static_cast<const u16*>(v);
  p.o = static_cast<u16*>(o);
  p.lse = lse;
  p.row_stride = row_stride;
  p.B = B; p.H = H; p.T = T;
  p.scale_log2 = softmax_scale * 1.4426950408889634f;
  // The wide kernel's workgroups take heads two at a time, so it has half as many work items; it wins (3-12 % at
  // B32-64 x H12 x T1024, B16 x H16 x T2048, B4 x H16 x T4096) once those fill its grid and the rows are long, ties
  // around that point (B22-24 x H12 x T1024, B96 x H12 x T512) and loses below (B8 x H12 x T1024: 39 vs 32 us,
  // B2 x H16 x T4096: 129 vs 107 us, B48 x H12 x T512: 55 vs 52 us).
  static const int env_mode = wide_forward_mode();  // 0 never, 1 when it pays, 2 whenever possible, 3: the 32-query 32x32x16 kernel
  auto launch_tile32 = [&]() {
    static const int per_cu32 = resident_per_cu(attn_fwd32_kernel);
    p.per_cu = per_cu32;
    hipLaunchKernelGGL(attn_fwd32_kernel, dim3(persistent_grid(per_cu32, T, B, H)), dim3(kThreads), 0, stream, p);
  };
  if (kernel == 2 || (kernel < 0 && env_mode == 3)) return launch_tile32();
  const int wide_mode = kernel == 0 ? 0 : kernel == 1 ? 2 : env_mode;
  if (wide_mode > 0 && (static_cast<long long>(B) * H) % 2 == 0) {
    static const int per_cu = wide_per_cu();
    const unsigned grid = persistent_grid(per_cu, T, B, H, 2);
    const long long items = static_cast<long long>((T / BQ + 1) / 2) * B * H / 2;
    if (wide_mode == 2 || (items >= static_cast<long long>(cu_count()) * per_cu && T >= 1024)) {
      p.per_cu = per_cu;
      hipLaunchKernelGGL(attn_fwd_wide_kernel, dim3(grid), dim3(kWideThreads), 0, stream, p);
      return;
    }
  }
  // (The 32-query kernel on 32x32x16 instructions measured 0-5 % ahead of the first kernel alone on every shape tried
  // and 0.7 % behind it inside a batch-16 GPT-2 step: selectable, not the default.)
  static const int fwd_per_cu = resident_per_cu(attn_fwd_kernel);
  p.per_cu = fwd_per_cu;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(persistent_grid(fwd_per_cu, T, B, H)), dim3(kThreads), 0, stream, p);
}

void launch_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta,
                          void* dq, void* dk, void* dv, long long row_stride, long long grad_row_stride, int B, int H, int T,
                          float softmax_scale, float* bias_partial, hipStream_t stream, int kernel) {
  AttnBwdParams p;
  p.o = static_cast<const u16*>(o);
  p.q = static_cast<const u16*>(q);
  p.k = static_cast<const u16*>(k);
  p.v = static_cast<const u16*>(v);
  p.d_o = static_cast<const u16*>(d_o);
  p.lse = lse;
  p.delta = delta;  // workspace of 2*B*H*T floats: delta, then lse in the log2 domain
  p.lse2 = delta + static_cast<long long>(B) * H * T;
  p.bias_partial = bias_partial;
  p.dq = static_cast<u16*>(dq);
  p.dk = static_cast<u16*>(dk);
  p.dv = static_cast<u16*>(dv);
  p.row_stride = row_stride;
  p.grad_row_stride = grad_row_stride;
  p.B = B; p.H = H; p.T = T;
  p.scale = softmax_scale;
  p.scale_log2 = softmax_scale * 1.4426950408889634f;
  // dQ first: it also produces delta, which the dK/dV kernel consumes
  static const int env_mode = backward_mode();
  const bool wide = kernel < 0 ? env_mode != 0 : kernel == 1;
  if (wide) {  // the kernels on 32x32x16 matrix instructions
    static const int dq_per_cu = resident_per_cu(attn_bwd_dq32_kernel), dkdv_per_cu = resident_per_cu(attn_bwd_dkdv32_kernel);
    p.per_cu_dq = dq_per_cu;
    p.per_cu_dkdv = dkdv_per_cu;
    hipLaunchKernelGGL(attn_bwd_dq32_kernel, dim3(persistent_grid(dq_per_cu, T, B, H)), dim3(kThreads), 0, stream, p);
    hipLaunchKernelGGL(attn_bwd_dkdv32_kernel, dim3(persistent_grid(dkdv_per_cu, T, B, H)), dim3(kThreads), 0, stream, p);
    return;
  }
  static const int dq_per_cu = resident_per_cu(attn_bwd_dq_kernel), dkdv_per_cu = resident_per_cu(attn_bwd_dkdv_kernel);
  p.per_cu_dq = dq_per_cu;
  p.per_cu_dkdv = dkdv_per_cu;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(persistent_grid(dq_per_cu, T, B, H)), dim3(kThreads), 0, stream, p);
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3(persistent_grid(dkdv_per_cu, T, B, H)), dim3(kThreads), 0, stream, p);
}

}  // namespace kernels
}  // namespace dpa
