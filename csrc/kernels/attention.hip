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
#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

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
};

// Loads this thread's two 16-byte pieces of a [64][64] tile (rows = keys key0..key0+63).
__device__ __forceinline__ void load_tile_regs(const u16* base, long long row_stride, int key0, u32x4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + i * kThreads;
    r[i] = *reinterpret_cast<const u32x4*>(base + static_cast<long long>(key0 + (id >> 3)) * row_stride + ((id & 7) << 3));
  }
}
template <bool IS_V>
__device__ __forceinline__ void store_tile_regs(char* tile, const u32x4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + i * kThreads;
    const int off = IS_V ? v_off(id >> 3, id & 7) : k_off(id >> 3, id & 7);
    *reinterpret_cast<u32x4*>(tile + off) = r[i];
  }
}

// Raw v_exp_f32 (2^x): inputs here are <= 0 or -inf, where the instruction is exact enough and
// needs none of the denormal-range fix-ups of exp2f().
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// One key tile for both 16-query strips of a wave.  MASK = the tile crosses the diagonal.
template <bool MASK>
__device__ __forceinline__ void fwd_tile(const char* ks, const char* vs, const bf16x8 (&qf)[2][2], f32x4 (&o)[2][4], float (&m)[2],
                                         float (&l)[2], int key0, int q0, int lane, float c) {
  const int g = lane >> 4, li = lane & 15;
  // K fragments (A operand): row = key 16t + li, d chunk 4*kk + g
  bf16x8 kf[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) kf[t][kk] = *reinterpret_cast<const bf16x8*>(ks + k_off(16 * t + li, 4 * kk + g));

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
    if (MASK) {
      const int rel = q0 + 16 * st + li - key0 - 4 * g;  // key index 16t + e is allowed while 16t + e <= rel
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (16 * t + e > rel) s[t][e] = -__builtin_huge_valf();
    }
    // row maximum on the raw scores (the scale is positive), then one fused multiply-add per element
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
    for (int t = 1; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m[st], mx * c);
    const float alpha = fast_exp2(m[st] - m_new);
    m[st] = m_new;
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = fast_exp2(fmaf(s[t][e], c, -m_new));
        s[t][e] = pe;
        ps += pe;
      }
    l[st] = fmaf(l[st], alpha, ps);  // per-lane partial sum; combined across the 4 lane groups at the end
    if (__any(alpha != 1.f)) {        // the running maximum moved for some query of this wave
#pragma unroll
      for (int td = 0; td < 4; ++td)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[st][td][e] *= alpha;
    }
#pragma unroll
    for (int sk = 0; sk < 2; ++sk)
      pf[st][sk] = pack8(s[2 * sk][0], s[2 * sk][1], s[2 * sk][2], s[2 * sk][3], s[2 * sk + 1][0], s[2 * sk + 1][1],
                         s[2 * sk + 1][2], s[2 * sk + 1][3]);
  }
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

__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * kTileBytes];  // [2 stages][K | V]
  const int nqb = p.T / BQ;
  const int qb = nqb - 1 - blockIdx.x;  // longest rows first
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int q0 = qb * BQ + wid * 32;

  const long long head_off = static_cast<long long>(b) * p.T * p.row_stride + static_cast<long long>(h) * HD;
  const u16* Q = p.q + head_off;
  const u16* K = p.k + head_off;
  const u16* V = p.v + head_off;

  // Q fragments (B operand): lane = query li of strip st, d = 32*kk + 8*g .. +7
  bf16x8 qf[2][2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qf[st][kk] = *reinterpret_cast<const bf16x8*>(Q + static_cast<long long>(q0 + 16 * st + li) * p.row_stride + 32 * kk + 8 * g);

  f32x4 o[2][4];
  float m[2], l[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    m[st] = -__builtin_huge_valf();
    l[st] = 0.f;
#pragma unroll
    for (int td = 0; td < 4; ++td) o[st][td] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int n_tiles = (qb * BQ + BQ) / BKV;
  u32x4 rk[2], rv[2];
  load_tile_regs(K, p.row_stride, 0, rk);
  load_tile_regs(V, p.row_stride, 0, rv);
  store_tile_regs<false>(smem, rk);
  store_tile_regs<true>(smem + kTileBytes, rv);
  __syncthreads();

  for (int jt = 0; jt < n_tiles; ++jt) {
    const int cur = jt & 1;
    const bool more = jt + 1 < n_tiles;
    if (more) {
      load_tile_regs(K, p.row_stride, (jt + 1) * BKV, rk);
      load_tile_regs(V, p.row_stride, (jt + 1) * BKV, rv);
    }
    const char* ks = smem + cur * 2 * kTileBytes;
    const char* vs = ks + kTileBytes;
    const int key0 = jt * BKV;

    if (key0 <= q0 + 31) {  // otherwise every key of the tile is in this wave's future
      if (key0 + BKV - 1 > q0) fwd_tile<true>(ks, vs, qf, o, m, l, key0, q0, lane, p.scale_log2);
      else fwd_tile<false>(ks, vs, qf, o, m, l, key0, q0, lane, p.scale_log2);
    }

    if (more) {
      char* kn = smem + (cur ^ 1) * 2 * kTileBytes;
      store_tile_regs<false>(kn, rk);
      store_tile_regs<true>(kn + kTileBytes, rv);
    }
    __syncthreads();
  }

  // Normalise and store.  Lane: query li of strip st, d = 16*td + 4*g + e.
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    float lt = l[st];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.f / lt;
    const int qi = q0 + 16 * st + li;
    u16* orow = p.o + (static_cast<long long>(b) * p.T + qi) * (static_cast<long long>(p.H) * HD) + static_cast<long long>(h) * HD;
#pragma unroll
    for (int td = 0; td < 4; ++td) {
      u16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = BF16::from_f32(o[st][td][e] * inv);
      *reinterpret_cast<u16x4*>(orow + 16 * td + 4 * g) = w;
    }
    if (g == 0) p.lse[(static_cast<long long>(b) * p.H + h) * p.T + qi] = (m[st] + log2f(lt)) * 0.6931471805599453f;
  }
}


// =====================================================================================
// Backward.  Three kernels, no atomics, no transposes through memory:
//   attn_bwd_delta   delta[b,h,t] = sum_d dO * O
//   attn_bwd_dq      same orientation as the forward (query on the lane): S^T = K Q^T and
//                    dP^T = V dO^T share K / V row fragments, lse and delta are per-lane scalars,
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
  const u16* d_o;     // [B, T, H, 64]
  const float* lse;   // [B, H, T]
  const float* delta; // [B, H, T]
  u16* dq;            // gradient tensors use grad_row_stride between consecutive t
  u16* dk;
  u16* dv;
  long long row_stride, grad_row_stride;
  int B, H, T;
  float scale, scale_log2;
};

__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const u16* __restrict__ o, const u16* __restrict__ d_o,
                                                             float* __restrict__ delta, int B, int H, int T) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;  // (b*T + t)*H + h
  if (idx >= static_cast<long long>(B) * T * H) return;
  const int h = static_cast<int>(idx % H);
  const long long bt = idx / H;
  const int t = static_cast<int>(bt % T), b = static_cast<int>(bt / T);
  const u16* po = o + idx * HD;
  const u16* pd = d_o + idx * HD;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < HD / 8; ++i) {
    float a[8], c[8];
    load_as_f32<BF16, 8>(po + 8 * i, a);
    load_as_f32<BF16, 8>(pd + 8 * i, c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(a[j], c[j], acc);
  }
  delta[(static_cast<long long>(b) * H + h) * T + t] = acc;
}

// Row fragment (A operand: row = tile row 16*t + li, k = 32*kk + 8*g ..) of a [rows][64] tile in the v_off image.
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int t, int kk, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + v_off(16 * t + (lane & 15), 4 * kk + (lane >> 4)));
}

template <bool MASK>
__device__ __forceinline__ void dq_tile(const char* ks, const char* vs, const bf16x8 (&qf)[2][2], const bf16x8 (&dof)[2][2],
                                        f32x4 (&dq)[2][4], const float (&lse2)[2], const float (&dl)[2], int key0, int q0, int lane,
                                        float c, float scale) {
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
        s[t][e] = pe * (dp[t][e] - dl[st]) * scale;
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

__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * kTileBytes];  // [2 stages][K | V]
  const int nqb = p.T / BQ;
  const int qb = nqb - 1 - blockIdx.x;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
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

  bf16x8 qf[2][2], dof[2][2];
  float lse2[2], dl[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int qi = q0 + 16 * st + li;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[st][kk] = *reinterpret_cast<const bf16x8*>(Q + static_cast<long long>(qi) * p.row_stride + 32 * kk + 8 * g);
      dof[st][kk] = *reinterpret_cast<const bf16x8*>(DO + static_cast<long long>(qi) * o_stride + 32 * kk + 8 * g);
    }
    lse2[st] = p.lse[stat_off + qi] * 1.4426950408889634f;
    dl[st] = p.delta[stat_off + qi];
  }
  f32x4 dq[2][4];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int td = 0; td < 4; ++td) dq[st][td] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_tiles = (qb * BQ + BQ) / BKV;
  u32x4 rk[2], rv[2];
  load_tile_regs(K, p.row_stride, 0, rk);
  load_tile_regs(V, p.row_stride, 0, rv);
  store_tile_regs<true>(smem, rk);
  store_tile_regs<true>(smem + kTileBytes, rv);
  __syncthreads();
  for (int jt = 0; jt < n_tiles; ++jt) {
    const int cur = jt & 1;
    const bool more = jt + 1 < n_tiles;
    if (more) {
      load_tile_regs(K, p.row_stride, (jt + 1) * BKV, rk);
      load_tile_regs(V, p.row_stride, (jt + 1) * BKV, rv);
    }
    const char* ks = smem + cur * 2 * kTileBytes;
    const char* vs = ks + kTileBytes;
    const int key0 = jt * BKV;
    if (key0 <= q0 + 31) {
      if (key0 + BKV - 1 > q0) dq_tile<true>(ks, vs, qf, dof, dq, lse2, dl, key0, q0, lane, p.scale_log2, p.scale);
      else dq_tile<false>(ks, vs, qf, dof, dq, lse2, dl, key0, q0, lane, p.scale_log2, p.scale);
    }
    if (more) {
      char* kn = smem + (cur ^ 1) * 2 * kTileBytes;
      store_tile_regs<true>(kn, rk);
      store_tile_regs<true>(kn + kTileBytes, rv);
    }
    __syncthreads();
  }
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int qi = q0 + 16 * st + li;
    u16* row = p.dq + (static_cast<long long>(b) * p.T + qi) * p.grad_row_stride + static_cast<long long>(h) * HD;
#pragma unroll
    for (int td = 0; td < 4; ++td) {
      u16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = BF16::from_f32(dq[st][td][e]);
      *reinterpret_cast<u16x4*>(row + 16 * td + 4 * g) = w;
    }
  }
}

constexpr int BQT = 32;                          // queries per tile in the dK/dV kernel
constexpr int kQTileBytes = BQT * HD * 2;        // 4 KiB

__device__ __forceinline__ void load_qtile_regs(const u16* base, long long row_stride, int row0, u32x4& r) {
  const int id = threadIdx.x;  // 256 pieces of 16 B: 32 rows x 8 chunks
  r = *reinterpret_cast<const u32x4*>(base + static_cast<long long>(row0 + (id >> 3)) * row_stride + ((id & 7) << 3));
}
__device__ __forceinline__ void store_qtile_regs(char* tile, const u32x4& r) {
  const int id = threadIdx.x;
  *reinterpret_cast<u32x4*>(tile + v_off(id >> 3, id & 7)) = r;
}

template <bool MASK>
__device__ __forceinline__ void dkdv_tile(const char* qs_, const char* dos_, const bf16x8 (&kfB)[2][2], const bf16x8 (&vfB)[2][2],
                                          f32x4 (&dk)[2][4], f32x4 (&dv)[2][4], const float* lse, const float* delta, int qt0,
                                          int k0w, int lane, float c, float scale) {
  const int g = lane >> 4, li = lane & 15;
  // per-register query statistics: query = qt0 + 16*qs + 4*g + e
  f32x4 l2[2], dl[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    l2[qs] = *reinterpret_cast<const f32x4*>(lse + qt0 + 16 * qs + 4 * g);
    dl[qs] = *reinterpret_cast<const f32x4*>(delta + qt0 + 16 * qs + 4 * g);
  }
  bf16x8 qrow[2][2], dorow[2][2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qrow[qs][kk] = row_frag(qs_, qs, kk, lane);
      dorow[qs][kk] = row_frag(dos_, qs, kk, lane);
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
        float pe = fast_exp2(fmaf(s[qs][e], c, -l2[qs][e] * 1.4426950408889634f));
        if (MASK && (16 * qs + e < rel)) pe = 0.f;
        pv[qs][e] = pe;
        dsv[qs][e] = pe * (dp[qs][e] - dl[qs][e]) * scale;
      }
    pB[kt2] = pack8(pv[0][0], pv[0][1], pv[0][2], pv[0][3], pv[1][0], pv[1][1], pv[1][2], pv[1][3]);
    dsB[kt2] = pack8(dsv[0][0], dsv[0][1], dsv[0][2], dsv[0][3], dsv[1][0], dsv[1][1], dsv[1][2], dsv[1][3]);
  }
  // dV^T[d][key] += dO^T P ; dK^T[d][key] += Q^T dS   (contraction over the tile's 32 queries)
#pragma unroll
  for (int td = 0; td < 4; ++td) {
    const bf16x8 dot = tr_frag(dos_, 0, td, lane);
    const bf16x8 qt = tr_frag(qs_, 0, td, lane);
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
      dv[kt2][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pB[kt2], dv[kt2][td], 0, 0, 0);
      dk[kt2][td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsB[kt2], dk[kt2][td], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dkdv_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * kQTileBytes];  // [2 stages][Q | dO]
  const int jb = blockIdx.x;  // key block; block 0 has the most query tiles and starts first
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
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
  const float* lse = p.lse + stat_off;
  const float* delta = p.delta + stat_off;

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
  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
    for (int td = 0; td < 4; ++td) {
      dk[kt2][td] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[kt2][td] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  const int first_qt = (jb * BQ) / BQT, n_qt = p.T / BQT;
  u32x4 rq, rdo;
  load_qtile_regs(Q, p.row_stride, first_qt * BQT, rq);
  load_qtile_regs(DO, o_stride, first_qt * BQT, rdo);
  store_qtile_regs(smem, rq);
  store_qtile_regs(smem + kQTileBytes, rdo);
  __syncthreads();
  for (int qt = first_qt; qt < n_qt; ++qt) {
    const int cur = (qt - first_qt) & 1;
    const bool more = qt + 1 < n_qt;
    if (more) {
      load_qtile_regs(Q, p.row_stride, (qt + 1) * BQT, rq);
      load_qtile_regs(DO, o_stride, (qt + 1) * BQT, rdo);
    }
    const char* qs_ = smem + cur * 2 * kQTileBytes;
    const char* dos_ = qs_ + kQTileBytes;
    const int qt0 = qt * BQT;
    if (qt0 + BQT - 1 >= k0w) {  // otherwise every query of the tile precedes all of this wave's keys
      if (qt0 < k0w + 31) dkdv_tile<true>(qs_, dos_, kfB, vfB, dk, dv, lse, delta, qt0, k0w, lane, p.scale_log2, p.scale);
      else dkdv_tile<false>(qs_, dos_, kfB, vfB, dk, dv, lse, delta, qt0, k0w, lane, p.scale_log2, p.scale);
    }
    if (more) {
      char* qn = smem + (cur ^ 1) * 2 * kQTileBytes;
      store_qtile_regs(qn, rq);
      store_qtile_regs(qn + kQTileBytes, rdo);
    }
    __syncthreads();
  }
  // lane: key = k0w + 16*kt2 + li, d = 16*td + 4*g + e
#pragma unroll
  for (int kt2 = 0; kt2 < 2; ++kt2) {
    const long long row = (static_cast<long long>(b) * p.T + k0w + 16 * kt2 + li) * p.grad_row_stride + static_cast<long long>(h) * HD;
#pragma unroll
    for (int td = 0; td < 4; ++td) {
      u16x4 wk, wv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wk[e] = BF16::from_f32(dk[kt2][td][e]);
        wv[e] = BF16::from_f32(dv[kt2][td][e]);
      }
      *reinterpret_cast<u16x4*>(p.dk + row + 16 * td + 4 * g) = wk;
      *reinterpret_cast<u16x4*>(p.dv + row + 16 * td + 4 * g) = wv;
    }
  }
}

}  // namespace

bool attention_supported(long long T, long long head_dim, long long row_stride) {
  return head_dim == HD && T % BQ == 0 && T >= BQ && row_stride % 8 == 0;
}

void launch_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, long long row_stride, int B, int H,
                          int T, float softmax_scale, hipStream_t stream) {
  AttnParams p;
  p.q = static_cast<const u16*>(q);
  p.k = static_cast<const u16*>(k);
  p.v = static_cast<const u16*>(v);
  p.o = static_cast<u16*>(o);
  p.lse = lse;
  p.row_stride = row_stride;
  p.B = B; p.H = H; p.T = T;
  p.scale_log2 = softmax_scale * 1.4426950408889634f;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(T / BQ, B * H), dim3(kThreads), 0, stream, p);
}

void launch_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta,
                          void* dq, void* dk, void* dv, long long row_stride, long long grad_row_stride, int B, int H, int T,
                          float softmax_scale, hipStream_t stream) {
  const long long n = static_cast<long long>(B) * T * H;
  hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream,
                     static_cast<const u16*>(o), static_cast<const u16*>(d_o), delta, B, H, T);
  AttnBwdParams p;
  p.q = static_cast<const u16*>(q);
  p.k = static_cast<const u16*>(k);
  p.v = static_cast<const u16*>(v);
  p.d_o = static_cast<const u16*>(d_o);
  p.lse = lse;
  p.delta = delta;
  p.dq = static_cast<u16*>(dq);
  p.dk = static_cast<u16*>(dk);
  p.dv = static_cast<u16*>(dv);
  p.row_stride = row_stride;
  p.grad_row_stride = grad_row_stride;
  p.B = B; p.H = H; p.T = T;
  p.scale = softmax_scale;
  p.scale_log2 = softmax_scale * 1.4426950408889634f;
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3(T / BQ, B * H), dim3(kThreads), 0, stream, p);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(T / BQ, B * H), dim3(kThreads), 0, stream, p);
}

}  // namespace kernels
}  // namespace dpa
