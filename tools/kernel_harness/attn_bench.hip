// Standalone timing harness for attention kernel experiments: includes the kernel source directly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include ATTN_SRC
int main(int argc, char** argv) {
  const int B = 32, H = 12, T = 1024;
  const long long C3 = 3LL * H * 64, n = 1LL * B * T * C3;
  std::vector<unsigned short> h(n);
  srand(1);
  for (auto& x : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
  unsigned short *qkv, *o, *d_o, *dqkv; float *lse, *delta;
  hipMalloc(&qkv, n * 2); hipMalloc(&o, n * 2 / 3); hipMalloc(&d_o, n * 2 / 3); hipMalloc(&dqkv, n * 2);
  hipMalloc(&lse, 4LL * B * H * T); hipMalloc(&delta, 8LL * B * H * T);
  hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_o, h.data(), n * 2 / 3, hipMemcpyHostToDevice);
  const long long Cd = H * 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto fwd = [&] { dpa::kernels::launch_attention_fwd(qkv, qkv + Cd, qkv + 2 * Cd, o, lse, C3, B, H, T, 0.125f, 0); };
  auto bwd = [&] { dpa::kernels::launch_attention_bwd(qkv, qkv + Cd, qkv + 2 * Cd, o, d_o, lse, delta, dqkv, dqkv + Cd, dqkv + 2 * Cd, C3, C3, B, H, T, 0.125f, nullptr, 0); };
  for (int i = 0; i < 3; ++i) { fwd(); bwd(); }
  float ms;
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) fwd(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("%s fwd %.1f us", argv[0], ms * 50);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) bwd(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("  bwd %.1f us\n", ms * 50);
  return 0;
}
