// Correctness screen + timing for the bf16 GEMM kernels, without torch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Icsrc -Icsrc/kernels -fno-math-errno -ffp-contract=fast \
//         -DGEMM_SRC='"gemm_bf16.hip"' [-DOLD_SRC='"/tmp/ab/old.hip"'] tools/kernel_harness/gemm_check.hip -o /tmp/gemm_check
//   gemm_check check            every layout x a list of shapes (incl. 1-3 k-steps, ragged edges, split-K),
//                               against an fp64 host reference (full for small products, sampled for
//                               large ones), each product run several times and compared bit for bit
//   gemm_check bench M N K L    L in NT NN TN TT; interleaved rounds, median / best TFLOP/s (random operands)
// OLD_SRC (optional) is a second copy of the kernel source whose dpa::kernels contents sit in a nested
// namespace old2ph; `bench` then alternates the two in one process.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include GEMM_SRC
#ifdef OLD_SRC
#include OLD_SRC
#endif

using dpa::kernels::GemmArgs;

static float bf2f(unsigned short h) { unsigned u = static_cast<unsigned>(h) << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return static_cast<unsigned short>(u >> 16); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Problem {
  long long M, N, K;
  int ak, bk;  // k-major flags
  std::vector<unsigned short> hA, hB, hBias;
  unsigned short *A = nullptr, *B = nullptr, *C = nullptr, *bias = nullptr;
  float* ws = nullptr;
  GemmArgs args{};
  Problem(long long M_, long long N_, long long K_, int ak_, int bk_, unsigned seed) : M(M_), N(N_), K(K_), ak(ak_), bk(bk_) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    hA.resize(M * K); hB.resize(N * K);
    for (auto& x : hA) x = f2bf(d(rng));
    for (auto& x : hB) x = f2bf(d(rng));
    CK(hipMalloc(&A, M * K * 2)); CK(hipMalloc(&B, N * K * 2)); CK(hipMalloc(&C, M * N * 2));
    CK(hipMemcpy(A, hA.data(), M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), N * K * 2, hipMemcpyHostToDevice));
    args.A = A; args.B = B; args.C = C; args.M = M; args.N = N; args.K = K;
    args.lda = ak ? K : M; args.ldb = bk ? K : N; args.ldc = N;
    args.a_kmajor = ak; args.b_kmajor = bk; args.alpha = 1.f;
  }
  ~Problem() { hipFree(A); hipFree(B); hipFree(C); if (ws) hipFree(ws); if (bias) hipFree(bias); }
  void add_bias() {
    std::mt19937 rng(4242);
    std::uniform_real_distribution<float> d(-4.f, 4.f);
    hBias.resize(N);
    for (auto& x : hBias) x = f2bf(d(rng));
    CK(hipMalloc(&bias, N * 2));
    CK(hipMemcpy(bias, hBias.data(), N * 2, hipMemcpyHostToDevice));
    args.bias = bias;
  }
  // element (m, k) of A / (n, k) of B as stored
  float a(long long m, long long k) const { return bf2f(ak ? hA[m * K + k] : hA[k * M + m]); }
  float b(long long n, long long k) const { return bf2f(bk ? hB[n * K + k] : hB[k * N + n]); }
  double ref(long long m, long long n) const { double s = hBias.empty() ? 0.0 : bf2f(hBias[n]); for (long long k = 0; k < K; ++k) s += static_cast<double>(a(m, k)) * b(n, k); return s; }
  void split(int s) { args.split_k = s; if (s > 1) { CK(hipMalloc(&ws, static_cast<size_t>(s) * M * N * 4)); args.workspace = ws; } }
};

// launches the 256x256 kernel whatever the shape (the dispatcher would send small products to the 128x128 kernel)
static void launch_large(const GemmArgs& a, hipStream_t st, bool streaming = false) {
  using namespace dpa::kernels;
  static unsigned* counter = nullptr;
  if (!counter) CK(hipMalloc(&counter, 32));  // one counter per XCD segment
  const bool split = a.split_k > 1 && a.workspace;
  const int tm = static_cast<int>((a.M + 255) / 256), tn = static_cast<int>((a.N + 255) / 256);
  const long long nk = a.K / 64;
  const int k_iters = split ? static_cast<int>((nk + a.split_k - 1) / a.split_k) : static_cast<int>(nk);
  streaming = streaming && !split && nk >= 3 && tm * tn >= 64;
  if (streaming) CK(hipMemsetAsync(counter, 0, 32, st));
  unsigned* ctr = streaming ? counter : nullptr;
  // streaming: the same number W of workgroups on every XCD, and 2 W ids in each of the eight segments of the id list
  const dim3 grid(streaming ? std::min(256, 4 * (tm * tn / 8)) & ~7 : tm * tn, split ? a.split_k : 1);
  const size_t lds = kLdsBytes2;
  auto go = [&](auto ak, auto bk) {
    constexpr bool AK = decltype(ak)::value, BKM = decltype(bk)::value;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_8ph_kernel<AK, BKM, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_8ph_kernel<AK, BKM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (split) {
      hipLaunchKernelGGL((gemm_bf16_8ph_kernel<AK, BKM, true>), grid, dim3(512), lds, st, a, tm, tn, k_iters, ctr);
      const long long quads = a.M * (a.N >> 2);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL((gemm_bf16_8ph_kernel<AK, BKM, false>), grid, dim3(512), lds, st, a, tm, tn, k_iters, ctr);
    }
  };
  if (a.a_kmajor && a.b_kmajor) go(std::true_type{}, std::true_type{});
  else if (a.a_kmajor) go(std::true_type{}, std::false_type{});
  else if (a.b_kmajor) go(std::false_type{}, std::true_type{});
  else go(std::false_type{}, std::false_type{});
}

static int check_one(long long M, long long N, long long K, int ak, int bk, int split, int reps, bool streaming = false, bool bias = false) {
  Problem p(M, N, K, ak, bk, 1234 + static_cast<unsigned>(M * 7 + N * 3 + K + ak * 2 + bk));
  p.split(split);
  if (bias) p.add_bias();
  std::vector<unsigned short> out0(M * N), out(M * N);
  int bad = 0, nondet = 0;
  for (int r = 0; r < reps; ++r) {
    CK(hipMemset(p.C, 0xff, M * N * 2));
    launch_large(p.args, 0, streaming);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), p.C, M * N * 2, hipMemcpyDeviceToHost));
    if (r == 0) out0 = out;
    else if (memcmp(out0.data(), out.data(), M * N * 2) != 0) ++nondet;
  }
  const bool full = M * N * K <= 300000000ll;
  std::mt19937 rng(99);
  const long long samples = full ? M * N : 6000;
  double worst = 0;
  for (long long s = 0; s < samples; ++s) {
    long long m, n;
    if (full) { m = s / N; n = s % N; }
    else if (s < 1500) { m = (s & 1) ? M - 1 - (rng() % std::min<long long>(M, 300)) : rng() % std::min<long long>(M, 300); n = rng() % N; }  // edges
    else { m = rng() % M; n = rng() % N; }
    const double ref = p.ref(m, n);
    const double got = bf2f(out0[m * N + n]);
    const double tol = 0.02 * std::sqrt(static_cast<double>(K)) * 0.35 + std::fabs(ref) * 0.01;  // bf16 rounding of the result + accumulation order
    const double err = std::fabs(got - ref);
    if (!(err <= tol)) { if (bad < 5) printf("    mismatch (%lld,%lld): got %.4f ref %.4f\n", m, n, got, ref); ++bad; }
    worst = std::max(worst, err / (std::fabs(ref) + 1.0));
  }
  printf("%s%s%s%s M=%lld N=%lld K=%lld split=%d : %s  (bad %d / %lld, worst rel %.3g, nondeterministic runs %d / %d)\n", streaming ? "streaming " : "", bias ? "bias " : "", ak ? "N" : "T", bk ? "T" : "N",
         M, N, K, split, (bad || nondet) ? "FAIL" : "ok", bad, samples, worst, nondet, reps - 1);
  fflush(stdout);
  return (bad || nondet) ? 1 : 0;
}

static double time_fn(const GemmArgs& a, int which, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) {
    if (which == 0) dpa::kernels::launch_gemm_bf16(a, 0);
#ifdef OLD_SRC
    else dpa::kernels::old2ph::launch_gemm_bf16(a, 0);
#endif
  }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / iters;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "check";
  if (mode == "check") {
    int fails = 0;
    const int layouts[4][2] = {{1, 1}, {1, 0}, {0, 0}, {0, 1}};
    struct S { long long M, N, K; int split; };
    const S shapes[] = {{256, 256, 64, 1},   {256, 256, 128, 1},  {512, 256, 192, 1},  {256, 512, 256, 1},  {512, 768, 320, 1},
                        {1024, 768, 768, 1}, {504, 520, 256, 1},  {264, 1000, 384, 1}, {768, 768, 2048, 4}, {768, 3072, 4096, 3},
                        {4096, 2304, 768, 1}, {8192, 768, 3072, 1}, {2048, 2048, 8192, 1}};
    for (const auto& l : layouts)
      for (const auto& s : shapes) fails += check_one(s.M, s.N, s.K, l[0], l[1], s.split, 4);
    // several tiles per workgroup (ids drawn from the counter): more tiles than CUs, 2 .. 48 k-steps, ragged edges
    const S stream_shapes[] = {{8192, 4096, 256, 1}, {16384, 2304, 192, 1}, {12288, 3072, 768, 1}, {8440, 2312, 256, 1}, {65536, 768, 3072, 1}, {2048, 1024, 512, 1}};
    for (const auto& l : layouts)
      for (const auto& s : stream_shapes) fails += check_one(s.M, s.N, s.K, l[0], l[1], 1, 4, true);
    // bias (carried in the accumulators of full tiles, added element-wise on ragged ones), both launch modes
    const S bias_shapes[] = {{512, 768, 320, 1}, {504, 520, 256, 1}, {768, 768, 2048, 4}, {12288, 3072, 768, 1}, {8440, 2312, 256, 1}};
    for (const auto& s : bias_shapes) {
      fails += check_one(s.M, s.N, s.K, 1, 1, s.split, 4, false, true);
      fails += check_one(s.M, s.N, s.K, 1, 0, s.split, 4, true, true);
    }
    printf(fails ? "FAILED: %d cases\n" : "all cases passed (%d failures)\n", fails);
    return fails ? 1 : 0;
  }
  if (mode == "bench") {
    const long long M = atoll(argv[2]), N = atoll(argv[3]), K = atoll(argv[4]);
    const std::string L = argc > 5 ? argv[5] : "NT";
    const int rounds = argc > 6 ? atoi(argv[6]) : 7;
    Problem p(M, N, K, L[0] == 'N', L[1] == 'T', 7);
    p.split(dpa::kernels::gemm_bf16_split_k(p.args));
    const double flop = 2.0 * M * N * K;
    const int iters = std::max(3, static_cast<int>(2e13 / flop));
    int nvar = 1;
#ifdef OLD_SRC
    nvar = 2;
#endif
    std::vector<double> tf[2];
    for (int v = 0; v < nvar; ++v) time_fn(p.args, v, 2);
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < nvar; ++v) tf[v].push_back(flop / (time_fn(p.args, v, iters) * 1e-3) / 1e12);
    for (int v = 0; v < nvar; ++v) {
      std::sort(tf[v].begin(), tf[v].end());
      printf("%s %lldx%lldx%lld split=%d %s: median %.0f TF  best %.0f TF  (%.1f us)\n", L.c_str(), M, N, K, p.args.split_k, v ? "old-2phase" : "new-8phase",
             tf[v][tf[v].size() / 2], tf[v].back(), flop / tf[v][tf[v].size() / 2] / 1e6);
    }
    return 0;
  }
#ifdef DPA_GEMM_STAMPS
  if (mode == "stamps") {  // shares of prologue / main loop / epilogue per workgroup, and the clock the chip holds
    const long long M = atoll(argv[2]), N = atoll(argv[3]), K = atoll(argv[4]);
    const std::string L = argc > 5 ? argv[5] : "NT";
    Problem p(M, N, K, L[0] == 'N', L[1] == 'T', 7);
    const long long nwg = ((M + 255) / 256) * ((N + 255) / 256);
    unsigned long long* st; CK(hipMalloc(&st, nwg * 64)); CK(hipMemset(st, 0, nwg * 64));
    p.args.colsum_partial = reinterpret_cast<float*>(st);
    const bool streaming = argc > 6 && std::string(argv[6]) == "stream";
    auto launch_large = [&](const GemmArgs& a, hipStream_t st) { ::launch_large(a, st, streaming); };
    for (int i = 0; i < 20; ++i) launch_large(p.args, 0);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); launch_large(p.args, 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nwg * 8);
    CK(hipMemcpy(h.data(), st, nwg * 64, hipMemcpyDeviceToHost));
    std::vector<double> d[3]; unsigned long long lo = ~0ull, hi = 0;
    for (long long w = 0; w < nwg; ++w) {
      for (int i = 0; i < 3; ++i) d[i].push_back(static_cast<double>(h[w * 8 + i + 1] - h[w * 8 + i]));
      lo = std::min(lo, h[w * 8]); hi = std::max(hi, h[w * 8 + 3]);
    }
    std::vector<double> clk; double busy = 0; unsigned long long rlo = ~0ull, rhi = 0;
    for (long long w = 0; w < nwg; ++w) {
      const double dc = static_cast<double>(h[w * 8 + 3] - h[w * 8]), dr = static_cast<double>(h[w * 8 + 5] - h[w * 8 + 4]);
      clk.push_back(dc / dr * 0.1);  // GHz: the real-time counter runs at 100 MHz
      busy += dr;
      rlo = std::min(rlo, h[w * 8 + 4]); rhi = std::max(rhi, h[w * 8 + 5]);
    }
    std::sort(clk.begin(), clk.end());
    printf("  in-kernel clock median %.2f GHz; workgroups resident %.1f %% of 256 CUs x (first entry -> last exit = %.1f us)\n", clk[clk.size() / 2],
           100.0 * busy / 256.0 / static_cast<double>(rhi - rlo), (rhi - rlo) * 0.01);
    for (auto& v : d) std::sort(v.begin(), v.end());
    printf("%s %lldx%lldx%lld: %.1f us wall; first stamp -> last stamp %.0f cycles => clock %.2f GHz\n", L.c_str(), M, N, K, ms * 1e3, static_cast<double>(hi - lo),
           (hi - lo) / (ms * 1e-3) / 1e9);
    const char* names[3] = {"entry -> first k-step landed", "main loop", "epilogue"};
    for (int i = 0; i < 3; ++i) printf("  %-30s median %8.0f  p10 %8.0f  p90 %8.0f cycles\n", names[i], d[i][d[i].size() / 2], d[i][d[i].size() / 10], d[i][d[i].size() * 9 / 10]);
    {  // last tile of every workgroup: stamps 6 (tile start), 2 (k-steps done), 3 (epilogue done)
      std::vector<double> kl, ep;
      for (long long w = 0; w < std::min<long long>(nwg, streaming ? 256 : nwg); ++w) {
        kl.push_back(static_cast<double>(h[w * 8 + 2] - h[w * 8 + 6]));
        ep.push_back(static_cast<double>(h[w * 8 + 3] - h[w * 8 + 2]));
      }
      std::sort(kl.begin(), kl.end()); std::sort(ep.begin(), ep.end());
      printf("  last tile per workgroup: k-steps median %.0f cycles (%.0f per k-step), epilogue median %.0f cycles\n", kl[kl.size() / 2], kl[kl.size() / 2] / (K / 64),
             ep[ep.size() / 2]);
    }
    printf("  k-steps %lld: main loop %.0f cycles per k-step (MFMA floor 2048)\n", K / 64, d[1][d[1].size() / 2] / (K / 64));
    return 0;
  }
#endif
  printf("usage: gemm_check check | bench M N K [NT|NN|TN|TT] [rounds]\n");
  return 1;
}
