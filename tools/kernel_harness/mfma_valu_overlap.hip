// Feasibility probe for a one-wave-per-SIMD attention body: how many cycles does ONE wave need per iteration for
// 16 independent v_mfma_f32_32x32x16_bf16 plus the softmax-like VALU work of a 64 x 32 score block (32 v_exp_f32,
// 32 v_fma_f32, 16 v_max3_f32, 16 v_cvt_pk_bf16_f32), (a) matrix only, (b) vector only, (c) both in one stream,
// interleaved 1 MFMA : 6 VALU by sched_group_barrier?  Full overlap would make (c) ~ max(a, b).
//   mfma_valu_overlap [waves_per_simd=1]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void body(float* out, unsigned long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = static_cast<__bf16>(0.001f * (lane + i));
    b[i] = static_cast<__bf16>(0.002f * (lane - i));
  }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  float s[32];
  for (int e = 0; e < 32; ++e) s[e] = 0.01f * (lane + e);
  float m = 0.5f, c = 1.1f, keep = 0.f;
  unsigned pk[16];
  for (int e = 0; e < 16; ++e) pk[e] = 0;
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
    }
    if (MODE != 0) {
      float mx = s[0];
#pragma unroll
      for (int e = 1; e < 32; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[e]), s[(e + 1) & 31]);   // 16 max3
      float p[32];
#pragma unroll
      for (int e = 0; e < 32; ++e) p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -m));             // 32 fma + 32 exp
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 v;
        v[0] = static_cast<__bf16>(p[2 * e]);
        v[1] = static_cast<__bf16>(p[2 * e + 1]);
        pk[e] ^= __builtin_bit_cast(unsigned, v);                                                           // 16 cvt_pk (+ xor)
      }
      keep += mx;
#pragma unroll
      for (int e = 0; e < 32; ++e) s[e] += 1e-6f * p[(e + 1) & 31];   // keeps the chain live without being foldable
    }
    if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);   // 8 VALU
      }
    }
  }
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float r = keep;
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) r += acc[j][e];
  for (int e = 0; e < 16; ++e) r += static_cast<float>(pk[e] & 0xff);
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  const int iters = 2000, blocks = 256 * wps;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  unsigned long long* h = new unsigned long long[blocks];
  const char* names[3] = {"16 MFMA 32x32x16 only", "softmax VALU only   ", "both, interleaved    "};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(body<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
      if (mode == 1) hipLaunchKernelGGL(body<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
      if (mode == 2) hipLaunchKernelGGL(body<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += h[i];
    printf("%s: %.1f cycles per iteration (waves per SIMD %d)\n", names[mode], sum / blocks / iters, wps);
  }
  return 0;
}
