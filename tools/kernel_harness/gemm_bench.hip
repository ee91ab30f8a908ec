#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include GEMM_SRC
// args: M N K [epilogue: 0 plain, 1 bias, 2 bias + GELU, 3 bias + GELU + pre-activation output, 4 activation backward]
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
  if (epi >= 1 && epi <= 3) { hipMalloc(&bias, N * 2); hipMemset(bias, 0, N * 2); a.bias = bias; }
  if (epi == 2 || epi == 3) a.act = dpa::kernels::kActGelu;
  if (epi == 3) { hipMalloc(&P, M * N * 2); a.pre_act = P; }
  if (epi == 4) {  // activation-backward epilogue: C = acc * GELU'(aux), one tile per workgroup
    hipMalloc(&P, M * N * 2);
    {  // pre-activations spread over +-4, so that the table look-ups of a wave go to different entries
      std::vector<unsigned short> hx(1 << 20);
      for (auto& x : hx) { float f = (rand() / (float)RAND_MAX - 0.5f) * 8.f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
      for (long long off = 0; off < M * N; off += (1 << 20))
        hipMemcpy(P + off, hx.data(), std::min<long long>(1 << 20, M * N - off) * 2, hipMemcpyHostToDevice);
    }
    a.aux = P; a.aux_mode = dpa::kernels::kActGelu;
    if (!dpa::kernels::gemm_bf16_act_backward_supported(a)) { printf("activation-backward epilogue: shape not supported\n"); return 1; }
  }
#ifdef DPA_GEMM_STAMPS
  unsigned long long* stamps;
  hipMalloc(&stamps, 4096 * 8 * 8); hipMemset(stamps, 0, 4096 * 8 * 8);
  a.colsum_partial = reinterpret_cast<float*>(stamps);
#endif
#ifdef GEMM_TILE_TIMES
  unsigned long long* times;
  const long long kTimeWgs = epi == 4 ? (M / 256) * (N / 256) : 256;         // workgroups that record
  const size_t kTimeBytes = static_cast<size_t>(kTimeWgs) * 128 * 8 * 8;   // [workgroup][tile < 128][slot < 8]
  hipMalloc(&times, kTimeBytes); hipMemset(times, 0, kTimeBytes);
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
  {
    hipMemset(times, 0, kTimeBytes);
    dpa::kernels::launch_gemm_bf16(a, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(kTimeBytes / 8);
    hipMemcpy(t.data(), times, kTimeBytes, hipMemcpyDeviceToHost);
    auto T = [&](long long wg, int tile, int slot) { return t[(wg * 128 + tile) * 8 + slot]; };
    if (epi == 4) {  // one tile per workgroup: stations of the activation-backward epilogue, mean over the workgroups
      const char* names[7] = {"entry -> tile start (table, offsets)", "tile start -> k-steps done", "k-steps done -> first aux half landed", "first half processed",
                              "second aux half landed", "first half stored + second processed", "second half stored"};
      const int order[8] = {7, 0, 2, 3, 4, 5, 6, 1};
      double sum[7] = {0}; long long n = 0;
      double whole = 0; long long nw = 0;  // later tiles of a workgroup that walks: start of one tile to start of the next
      for (long long w = 0; w < kTimeWgs; ++w) {
        if (!T(w, 0, 1)) continue;
        for (int i = 0; i < 7; ++i) sum[i] += double(T(w, 0, order[i + 1]) - T(w, 0, order[i]));
        ++n;
        for (int k = 1; k < 127 && T(w, k + 1, 0); ++k) {
          for (int i = 1; i < 7; ++i) sum[i] += double(T(w, k, order[i + 1]) - T(w, k, order[i]));
          whole += double(T(w, k + 1, 0) - T(w, k, 0));
          ++nw;
        }
      }
      printf("  first tiles %lld, later tiles %lld (%.2f us from tile start to tile start)\n", n, nw, nw ? whole / nw / 100.0 : 0.0);
      for (int i = 0; i < 7; ++i) printf("  %-40s %6.2f us\n", names[i], sum[i] / (i ? n + nw : n) / 100.0);
      return 0;
    }
    // where in its tile period is every workgroup when workgroup 0 ends tile k?  (spread = phases differ; one pile = in step)
    int ntile = 0; while (ntile < 128 && T(0, ntile, 1)) ++ntile;
    const double period = double(T(0, ntile - 1, 1) - T(0, 0, 0)) / ntile;
    printf("  workgroup 0: %d tiles, %.2f us per tile, of which k-steps %.2f us\n", ntile, period / 100.0,
           [&] { double s = 0; int n = 0; for (int w = 0; w < 256; ++w) for (int k = 0; k < 128 && T(w, k, 1); ++k) { s += double(T(w, k, 2) - T(w, k, 0)); ++n; } return n ? s / n / 100.0 : 0.0; }());
    for (int k : {1, 5, 10, 20, 40, 60}) {
      if (k >= ntile) break;
      const unsigned long long ref = T(0, k, 1);
      int hist[8] = {0};
      for (int w = 0; w < 256; ++w) {
        long long best = 1ll << 60;  // the end of a tile of workgroup w nearest to ref
        for (int j = 0; j < 128 && T(w, j, 1); ++j) { const long long d = (long long)(T(w, j, 1) - ref); if (llabs(d) < llabs(best)) best = d; }
        double ph = best / period + 0.5;  // 0 .. 1 around the reference
        if (ph < 0) ph = 0; if (ph > 0.999) ph = 0.999;
        ++hist[int(ph * 8)];
      }
      printf("  around tile %2d of workgroup 0, tile ends by eighth of a period (4 = together with it):", k);
      for (int h = 0; h < 8; ++h) printf(" %3d", hist[h]);
      printf("\n");
    }
  }
#endif
  return 0;
}
