#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include GEMM_SRC
int main(int argc, char** argv) {
  const long long M = argc > 1 ? atoll(argv[1]) : 8192, N = argc > 2 ? atoll(argv[2]) : 8192, K = argc > 3 ? atoll(argv[3]) : 8192;
  std::vector<unsigned short> h(M * K > N * K ? M * K : N * K);
  srand(1);
  for (auto& x : h) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
  unsigned short *A, *B, *C;
  hipMalloc(&A, M * K * 2); hipMalloc(&B, N * K * 2); hipMalloc(&C, M * N * 2);
  hipMemcpy(A, h.data(), M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), N * K * 2, hipMemcpyHostToDevice);
  dpa::kernels::GemmArgs a{};
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N; a.a_kmajor = 1; a.b_kmajor = 1; a.alpha = 1.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) dpa::kernels::launch_gemm_bf16(a, 0);
  float ms;
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) dpa::kernels::launch_gemm_bf16(a, 0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("%s %lldx%lldx%lld  %.1f us  %.0f TF\n", argv[0], M, N, K, ms * 100, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
  return 0;
}
