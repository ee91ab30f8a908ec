// Device-side helpers shared by the gfx950 kernels: 16-bit float conversion, 16-byte
// vector access, 64-lane wave reductions.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace dpa {
namespace kernels {

constexpr int kWave = 64;  // CDNA wavefront width

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct F32 {
  using T = float;
  static constexpr int kVec = 4;  // elements per 16-byte access
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
struct BF16 {
  using T = u16;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float to_f32(u16 v) {
    return __uint_as_float(static_cast<unsigned>(v) << 16);
  }
  static __device__ __forceinline__ u16 from_f32(float v) {
    __bf16 b = static_cast<__bf16>(v);  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(u16, b);
  }
};
struct F16 {
  using T = u16;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float to_f32(u16 v) {
    return static_cast<float>(__builtin_bit_cast(_Float16, v));
  }
  static __device__ __forceinline__ u16 from_f32(float v) {
    return __builtin_bit_cast(u16, static_cast<_Float16>(v));
  }
};

// Loads/stores N consecutive elements as fp32.  `aligned16` selects 16-byte accesses.
template <typename TR, int N>
__device__ __forceinline__ void load_as_f32(const typename TR::T* p, float (&out)[N]) {
  static_assert(N % TR::kVec == 0, "N must be a multiple of the 16-byte vector width");
  if constexpr (TR::kVec == 4) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[4 * i + j] = v[j];
    }
  } else {
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
      u16x8 v = *reinterpret_cast<const u16x8*>(p + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) out[8 * i + j] = TR::to_f32(v[j]);
    }
  }
}

template <typename TR, int N, bool STREAM = false>
__device__ __forceinline__ void store_from_f32(typename TR::T* p, const float (&in)[N]) {
  static_assert(N % TR::kVec == 0, "N must be a multiple of the 16-byte vector width");
  if constexpr (TR::kVec == 4) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = in[4 * i + j];
      if (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p + 4 * i));
      else *reinterpret_cast<f32x4*>(p + 4 * i) = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
      u16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = TR::from_f32(in[8 * i + j]);
      if (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<u16x8*>(p + 8 * i));
      else *reinterpret_cast<u16x8*>(p + 8 * i) = v;
    }
  }
}

// A few consecutive elements of a row as they sit in memory (16 or 8 bytes per lane), and their
// expansion to fp32: lets a kernel request data early and keep it packed (half the registers for the
// 16-bit types) until it is consumed.  V = elements per lane and access.
typedef __attribute__((ext_vector_type(4))) unsigned int raw16;
typedef __attribute__((ext_vector_type(2))) unsigned int raw8;
template <int BYTES>
struct RawVec;
template <>
struct RawVec<16> {
  using type = raw16;
  static constexpr int kDwords = 4;
};
template <>
struct RawVec<8> {
  using type = raw8;
  static constexpr int kDwords = 2;
};
template <typename TR, int V>
using raw_t = typename RawVec<V * static_cast<int>(sizeof(typename TR::T))>::type;

template <typename TR, int V>
__device__ __forceinline__ raw_t<TR, V> load_raw(const typename TR::T* p) {
  return *reinterpret_cast<const raw_t<TR, V>*>(p);
}
// Streaming variants: data that this kernel touches once and nobody reads again soon (non-temporal hint).
template <typename TR, int V>
__device__ __forceinline__ raw_t<TR, V> load_raw_stream(const typename TR::T* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const raw_t<TR, V>*>(p));
}
template <typename TR, int V>
__device__ __forceinline__ raw_t<TR, V> zero_raw() {
  raw_t<TR, V> r;
#pragma unroll
  for (int j = 0; j < RawVec<V * static_cast<int>(sizeof(typename TR::T))>::kDwords; ++j) r[j] = 0u;
  return r;
}
template <typename TR, int V>
__device__ __forceinline__ void raw_to_f32(const raw_t<TR, V>& r, float (&out)[V]) {
  if constexpr (sizeof(typename TR::T) == 4) {
#pragma unroll
    for (int j = 0; j < V; ++j) out[j] = __uint_as_float(r[j]);
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) out[j] = TR::to_f32(static_cast<u16>(j & 1 ? r[j >> 1] >> 16 : r[j >> 1] & 0xffffu));
  }
}
template <typename TR, int V, bool STREAM = false>
__device__ __forceinline__ void store_raw_from_f32(typename TR::T* p, const float (&in)[V]) {
  raw_t<TR, V> r;
  if constexpr (sizeof(typename TR::T) == 4) {
#pragma unroll
    for (int j = 0; j < V; ++j) r[j] = __float_as_uint(in[j]);
  } else {
#pragma unroll
    for (int j = 0; j < V / 2; ++j)
      r[j] = static_cast<unsigned>(TR::from_f32(in[2 * j])) | (static_cast<unsigned>(TR::from_f32(in[2 * j + 1])) << 16);
  }
  if (STREAM) __builtin_nontemporal_store(r, reinterpret_cast<raw_t<TR, V>*>(p));
  else *reinterpret_cast<raw_t<TR, V>*>(p) = r;
}

// Stops the compiler from expanding a loop-invariant packed vector to fp32 outside the loop (which
// would double its register footprint for the whole kernel); costs no instruction.
__device__ __forceinline__ void keep_packed(raw16& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void keep_packed(raw8& r) { asm volatile("" : "+v"(r)); }

// Wave-wide reductions on the data-parallel-primitive path of the VALU (the lane exchange is an operand
// modifier of the add itself, ~8 cycles per step) instead of six dependent ds_bpermute round trips through
// the LDS crossbar (~100 cycles each): row-of-16 butterflies, then the row totals are chained across the
// four rows with row_bcast and lane 63 is read back as a scalar.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppRowHalfMirror = 0x141, kDppRowMirror = 0x140, kDppRowBcast15 = 0x142,
              kDppRowBcast31 = 0x143;
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<kDppQuadXor1, 0xF>(v, 0.f);
  v += dpp_move<kDppQuadXor2, 0xF>(v, 0.f);
  v += dpp_move<kDppRowHalfMirror, 0xF>(v, 0.f);
  v += dpp_move<kDppRowMirror, 0xF>(v, 0.f);
  v += dpp_move<kDppRowBcast15, 0xA>(v, 0.f);  // rows 1 and 3 += total of the row before
  v += dpp_move<kDppRowBcast31, 0xC>(v, 0.f);  // rows 2 and 3 += lane 31 (rows 0 + 1)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  constexpr float kLow = -3.402823466e38f;
  v = fmaxf(v, dpp_move<kDppQuadXor1, 0xF>(v, kLow));
  v = fmaxf(v, dpp_move<kDppQuadXor2, 0xF>(v, kLow));
  v = fmaxf(v, dpp_move<kDppRowHalfMirror, 0xF>(v, kLow));
  v = fmaxf(v, dpp_move<kDppRowMirror, 0xF>(v, kLow));
  v = fmaxf(v, dpp_move<kDppRowBcast15, 0xA>(v, kLow));
  v = fmaxf(v, dpp_move<kDppRowBcast31, 0xC>(v, kLow));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sum across a workgroup of NWAVES waves; every thread gets the result.  `scratch` holds
// at least NWAVES floats of LDS and may be reused after the call returns.
template <int NWAVES>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  if constexpr (NWAVES == 1) return v;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NWAVES; ++i) r += scratch[i];
  return r;
}

template <typename F>
inline void dispatch_dtype(int dt, F&& f) {
  switch (dt) {
    case 0: f(F32{}); break;
    case 1: f(BF16{}); break;
    default: f(F16{}); break;
  }
}

// d/dx [x * Phi(x)] = Phi(x) + x * phi(x).  Phi comes from the Abramowitz-Stegun 7.1.26 rational
// form of erf (absolute error 1.5e-7, fp32-level), which needs exp(-x^2/2) -- the same exponential
// the density term uses -- so the whole derivative costs one v_exp, one v_rcp and eight FMAs
// instead of a libm erff call (the backward of the GELU layers is VALU-bound otherwise).
__device__ __forceinline__ float gelu_grad(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);  // exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, ax, 1.f));       // 1 / (1 + p*|x|/sqrt(2)), p = 0.3275911
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float tail = 0.5f * poly * t * e;            // 1 - Phi(|x|)
  const float cdf = x >= 0.f ? 1.f - tail : tail;
  return fmaf(x * 0.3989422804014327f, e, cdf);
}

// Phi(x), the standard normal distribution function, in the same rational erfc form.
__device__ __forceinline__ float normal_cdf(float x) {
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float tail = 0.5f * poly * t * e;
  return x >= 0.f ? 1.f - tail : tail;
}

// GELU(x) = x * Phi(x) with the same rational erfc form as gelu_grad: branch-free, one v_exp and one
// v_rcp (libm's erff costs several divergent branches per element, which an MFMA GEMM epilogue that
// runs 128 elements per lane cannot hide).
__device__ __forceinline__ float gelu_fwd(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, ax, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float tail = 0.5f * poly * t * e;  // 1 - Phi(|x|)
  return x * (x >= 0.f ? 1.f - tail : tail);
}

}  // namespace kernels
}  // namespace dpa
