#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include GEMM_SRC
// args: M N K [epilogue: 0 plain, 1 bias, 2 bias + GELU, 3 bias + GELU + pre-activation output]
// With -DDPA_GEMM_STAMPS: cycles of the last tile's k-loop and epilogue, averaged over the workgroups.
int main(int argc, char** argv) {
  const long long M = argc > 1 ? atoll(argv[1]) : 8192, N = argc > 2 ? atoll(argv[2]) : 8192, K = argc > 3 ? atoll(argv[3]) : 8192;
  const int epi = argc > 4 ? atoi(argv[4]) : 0;
  std::vector<unsigned short> h(M * K > N * K ? M * K : N * K);
  srand(1);
  for (auto& x : h) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
  unsigned short *A, *B, *C, *P = nullptr, *bias = nullptr;
  hipMalloc(&A, M * K * 2); hipMalloc(&B, N * K * 2); hipMalloc(&C, M * N * 2);
  hipMemcpy(A, h.data(), M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), N * K * 2, hipMemcpyHostToDevice);
  dpa::kernels::GemmArgs a{};
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N; a.a_kmajor = 1; a.b_kmajor = 1; a.alpha = 1.f;
  if (epi >= 1) { hipMalloc(&bias, N * 2); hipMemset(bias, 0, N * 2); a.bias = bias; }
  if (epi >= 2) a.act = dpa::kernels::kActGelu;
  if (epi >= 3) { hipMalloc(&P, M * N * 2); a.pre_act = P; }
#ifdef DPA_GEMM_STAMPS
  unsigned long long* stamps;
  hipMalloc(&stamps, 4096 * 8 * 8); hipMemset(stamps, 0, 4096 * 8 * 8);
  a.colsum_partial = reinterpret_cast<float*>(stamps);
#endif
#ifdef GEMM_TILE_TIMES
  unsigned long long* times;
  hipMalloc(&times, 256 * 128 * 2 * 8); hipMemset(times, 0, 256 * 128 * 2 * 8);
  a.colsum_partial = reinterpret_cast<float*>(times);
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) dpa::kernels::launch_gemm_bf16(a, 0);
  float ms;
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) dpa::kernels::launch_gemm_bf16(a, 0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("%s %lldx%lldx%lld epilogue %d  %.1f us  %.0f TF\n", argv[0], M, N, K, epi, ms * 100, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
#ifdef DPA_GEMM_STAMPS
  std::vector<unsigned long long> s(4096 * 8);
  hipMemcpy(s.data(), stamps, s.size() * 8, hipMemcpyDeviceToHost);
  double loop = 0, ep = 0, life = 0; int n = 0;
  for (int b = 0; b < 4096; ++b) {
    const unsigned long long* t = &s[b * 8];
    if (!t[3] || !t[6]) continue;
    loop += double(t[2] - t[6]); ep += double(t[3] - t[2]); life += double(t[5] - t[4]); ++n;
  }
  if (n) printf("  %d workgroups: last tile k-loop %.0f cycles, epilogue (until its last store is issued) %.0f cycles; lifetime %.1f us (100 MHz clock)\n",
                n, loop / n, ep / n, life / n / 100.0);
#endif
#ifdef GEMM_TILE_TIMES
  {  // where in its tile period is every workgroup when workgroup 0 starts tile k?  (spread = phases differ; one pile = in step)
    hipMemset(times, 0, 256 * 128 * 2 * 8);
    dpa::kernels::launch_gemm_bf16(a, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(256 * 128 * 2);
    hipMemcpy(t.data(), times, t.size() * 8, hipMemcpyDeviceToHost);
    auto T = [&](int wg, int tile, int slot) { return t[(wg * 128 + tile) * 2 + slot]; };
    int ntile = 0; while (ntile < 128 && T(0, ntile, 1)) ++ntile;
    const double period = double(T(0, ntile - 1, 1) - T(0, 0, 0)) / ntile;
    printf("  workgroup 0: %d tiles, %.2f us per tile; epilogue (first convert -> last store issued) mean %.2f us\n", ntile, period / 100.0,
           [&] { double s = 0; int n = 0; for (int w = 0; w < 256; ++w) for (int k = 0; k < 128 && T(w, k, 1); ++k) { s += double(T(w, k, 1) - T(w, k, 0)); ++n; } return n ? s / n / 100.0 : 0.0; }());
    printf("  (slot 0 is the tile start, slot 1 the end of its epilogue: the difference above is the whole tile)\n");
    for (int k : {1, 5, 10, 20, 40, 60}) {
      if (k >= ntile) break;
      const unsigned long long ref = T(0, k, 1);
      int hist[8] = {0};
      for (int w = 0; w < 256; ++w) {
        // the end of the epilogue of workgroup w nearest to ref
        long long best = 1ll << 60;
        for (int j = 0; j < 128 && T(w, j, 1); ++j) { const long long d = (long long)(T(w, j, 1) - ref); if (llabs(d) < llabs(best)) best = d; }
        double ph = best / period + 0.5;  // 0 .. 1 around the reference
        if (ph < 0) ph = 0; if (ph > 0.999) ph = 0.999;
        ++hist[int(ph * 8)];
      }
      printf("  around tile %2d of workgroup 0, epilogue ends by eighth of a period (4 = together with it):", k);
      for (int h = 0; h < 8; ++h) printf(" %3d", hist[h]);
      printf("\n");
    }
  }
#endif
  return 0;
}
