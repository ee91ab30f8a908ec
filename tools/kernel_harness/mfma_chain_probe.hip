// How long does one wave need for the matrix + vector work of one half-step of the one-wave-per-SIMD attention forward:
// 8 x v_mfma_f32_32x32x16_bf16 and 48 vector instructions (16 fma, 16 exp, 8 convert, 8 dot2), interleaved 1 : 6,
//   CHAIN = 1: four of the matrix instructions accumulate into ONE register block back to back (a score product),
//              the other four into two blocks alternately (an output product) -- the kernel's pattern;
//   CHAIN = 0: every matrix instruction has its own accumulator.
// and the same without the vector work?  Prints cycles per iteration for one wave per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(4))) float f32x4;

// The same FLOPs as body<0, VALU> from 16 x v_mfma_f32_16x16x32_bf16 (own accumulators), interleaved 1 : 3.
template <int VALU>
__global__ __launch_bounds__(256) void body16(float* out, unsigned long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = static_cast<__bf16>(0.001f * (lane + i));
    b[i] = static_cast<__bf16>(0.002f * (lane - i));
  }
  f32x4 acc[16];
  for (int j = 0; j < 16; ++j)
    for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
  float s[16];
  for (int e = 0; e < 16; ++e) s[e] = 0.01f * (lane + e);
  float m = 0.5f, c = 1.1f, lp = 0.f;
  bf16x2 ones;
  ones[0] = static_cast<__bf16>(1.f);
  ones[1] = static_cast<__bf16>(1.f);
  unsigned keep = 0;
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    if (VALU) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        bf16x2 v;
        v[0] = static_cast<__bf16>(__builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -m)));
        v[1] = static_cast<__bf16>(__builtin_amdgcn_exp2f(__builtin_fmaf(s[e + 1], c, -m)));
        lp = __builtin_amdgcn_fdot2_f32_bf16(v, ones, lp, false);
        keep ^= __builtin_bit_cast(unsigned, v);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
      }
      asm volatile("" : "+v"(s[it & 15]));
    }
  }
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float r = lp + static_cast<float>(keep & 0xff);
  for (int j = 0; j < 16; ++j)
    for (int e = 0; e < 4; ++e) r += acc[j][e];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int CHAIN, int VALU>
__global__ __launch_bounds__(256) void body(float* out, unsigned long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = static_cast<__bf16>(0.001f * (lane + i));
    b[i] = static_cast<__bf16>(0.002f * (lane - i));
  }
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  float s[16];
  for (int e = 0; e < 16; ++e) s[e] = 0.01f * (lane + e);
  float m = 0.5f, c = 1.1f, lp = 0.f;
  bf16x2 ones;
  ones[0] = static_cast<__bf16>(1.f);
  ones[1] = static_cast<__bf16>(1.f);
  unsigned keep = 0;
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int j = CHAIN ? (k < 4 ? 0 : 1 + (k & 1)) : k;
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    if (VALU) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        bf16x2 v;
        v[0] = static_cast<__bf16>(__builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -m)));
        v[1] = static_cast<__bf16>(__builtin_amdgcn_exp2f(__builtin_fmaf(s[e + 1], c, -m)));
        lp = __builtin_amdgcn_fdot2_f32_bf16(v, ones, lp, false);
        keep ^= __builtin_bit_cast(unsigned, v);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
      }
      asm volatile("" : "+v"(s[it & 15]));
    }
  }
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float r = lp + static_cast<float>(keep & 0xff);
  for (int j = 0; j < 8; ++j)
    for (int e = 0; e < 16; ++e) r += acc[j][e];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int CHAIN, int VALU>
void run(const char* name, float* out, unsigned long long* cyc, unsigned long long* h) {
  const int iters = 4000, blocks = 256;
  for (int rep = 0; rep < 2; ++rep) {
    if (CHAIN == 2) hipLaunchKernelGGL((body16<VALU>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    else hipLaunchKernelGGL((body<CHAIN == 2 ? 0 : CHAIN, VALU>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < blocks; ++i) sum += h[i];
  printf("%s: %.1f cycles per iteration\n", name, sum / blocks / iters);
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * sizeof(float));
  (void)hipMalloc(&cyc, 256 * sizeof(unsigned long long));
  unsigned long long h[256];
  run<0, 0>("8 matrix, own accumulators, no vector work     ", out, cyc, h);
  run<1, 0>("8 matrix, chain of 4 + 2 x 2, no vector work    ", out, cyc, h);
  run<0, 1>("8 matrix, own accumulators, 48 vector interleaved", out, cyc, h);
  run<1, 1>("8 matrix, chain of 4 + 2 x 2, 48 vector interleaved", out, cyc, h);
  run<2, 0>("16 matrix 16x16x32 (same FLOPs), no vector work      ", out, cyc, h);
  run<2, 1>("16 matrix 16x16x32 (same FLOPs), 48 vector interleaved", out, cyc, h);
  return 0;
}
