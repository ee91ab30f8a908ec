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
#ifdef ATTN_STAMP
  {
    // diagnostic build: stamps of the persistent forward kernel (wave 3 of every workgroup)
    static unsigned long long hs[1 << 12][8];
    hipMemset(hs, 0, 0);
    fwd();
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(dpa::kernels::g_attn_stamps), sizeof(hs));
    int nwg = 0;
    double loop = 0, tiles = 0, life = 0, sync = 0, life_rt = 0;
    unsigned long long t_min = ~0ull, t_max = 0;
    for (int i = 0; i < (1 << 12); ++i) {
      if (hs[i][1] == 0) continue;
      ++nwg;
      loop += hs[i][0]; tiles += hs[i][1]; life += hs[i][4]; sync += hs[i][5];
      life_rt += double(hs[i][3] - hs[i][2]);
      if (hs[i][2] < t_min) t_min = hs[i][2];
      if (hs[i][3] > t_max) t_max = hs[i][3];
    }
    printf("  %d workgroups: loop %.0f cyc per tile (of which waiting at the barrier %.0f), %.0f tiles each; loop is %.1f %% of a workgroup's life; span %.1f us\n",
           nwg, loop / tiles, sync / tiles, tiles / nwg, 100.0 * loop / life, double(t_max - t_min) * 1e-2);
    printf("  s_memtime ticks per microsecond of s_memrealtime (100 MHz): %.0f\n", life / (life_rt * 1e-2));
    int alive[16] = {0};
    for (int i = 0; i < (1 << 12); ++i) {
      if (hs[i][1] == 0) continue;
      for (int s16 = 0; s16 < 16; ++s16) {
        const unsigned long long t = t_min + (t_max - t_min) * (2 * s16 + 1) / 32;
        if (hs[i][2] <= t && t < hs[i][3]) alive[s16]++;
      }
    }
    {
      double by_xcd[8] = {0}, by_pair[4] = {0}, by_half[2] = {0};
      int n_x[8] = {0}, n_p[4] = {0}, n_h[2] = {0};
      for (int i = 0; i < (1 << 12); ++i) {
        if (hs[i][1] == 0) continue;
        const double us = double(hs[i][3] - hs[i][2]) * 1e-2;
        by_xcd[i & 7] += us; n_x[i & 7]++;
        by_pair[(i >> 3) & 3] += us; n_p[(i >> 3) & 3]++;
        by_half[i >= nwg / 2] += us; n_h[i >= nwg / 2]++;
      }
      printf("  life (us) by id&7:");
      for (int k = 0; k < 8; ++k) printf(" %.1f", by_xcd[k] / (n_x[k] ? n_x[k] : 1));
      printf(" | by (id>>3)&3:");
      for (int k = 0; k < 4; ++k) printf(" %.1f", by_pair[k] / (n_p[k] ? n_p[k] : 1));
      printf(" | first/second half of ids: %.1f %.1f\n", by_half[0] / (n_h[0] ? n_h[0] : 1), by_half[1] / (n_h[1] ? n_h[1] : 1));
    }
    printf("  alive workgroups along the span:");
    for (int s16 = 0; s16 < 16; ++s16) printf(" %d", alive[s16]);
    printf("\n");
  }
#endif
  return 0;
}
