// Where does a step of the one-wave-per-SIMD attention forward spend its cycles?  Compiles the production kernel file
// with WIDE_STAMP (s_memtime stamps around the sections of a loop step; they serialise the sections, so the total
// is longer than in production -- the split is what matters) and prints per-wave averages.
//   attn_wide_stamps [B=64] [H=12] [T=1024]
#define WIDE_STAMP 1
#include "../../csrc/kernels/attention.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 12, T = argc > 3 ? atoi(argv[3]) : 1024;
  const size_t n_qkv = static_cast<size_t>(B) * T * 3 * H * 64;
  std::vector<unsigned short> h(n_qkv);
  unsigned x = 12345;
  for (auto& v : h) {  // bf16 values in about [-2, 2)
    x = x * 1664525u + 1013904223u;
    const float f = (static_cast<int>(x >> 8) % 4096 - 2048) / 1024.f;
    unsigned bits;
    __builtin_memcpy(&bits, &f, 4);
    v = static_cast<unsigned short>(bits >> 16);
  }
  unsigned short *qkv, *o;
  float* lse;
  if (hipMalloc(&qkv, n_qkv * 2) != hipSuccess || hipMalloc(&o, n_qkv / 3 * 2) != hipSuccess ||
      hipMalloc(&lse, static_cast<size_t>(B) * H * T * 4) != hipSuccess) return 1;
  (void)hipMemcpy(qkv, h.data(), n_qkv * 2, hipMemcpyHostToDevice);
  const long long rs = 3LL * H * 64;
  for (int rep = 0; rep < 3; ++rep)
    dpa::kernels::launch_attention_fwd(qkv, qkv + H * 64, qkv + 2 * H * 64, o, lse, rs, B, H, T, 0.125f, nullptr);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, nullptr);
  dpa::kernels::launch_attention_fwd(qkv, qkv + H * 64, qkv + 2 * H * 64, o, lse, rs, B, H, T, 0.125f, nullptr);
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long st[1024][20];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(dpa::kernels::g_wide_stamps), sizeof st);
  const char* names[7] = {"counted DMA wait", "barrier", "DMA requests", "half 0", "between (qb 1)", "half 1", "between (qb 0)"};
  printf("kernel %.1f us with stamps\n", ms * 1e3);
  for (int w = 0; w < 4; ++w) {
    double acc[20] = {0};
    for (int blk = 0; blk < 256; ++blk)
      for (int i = 0; i < 20; ++i) acc[i] += st[blk * 4 + w][i];
    printf("wave %d: %.0f full steps per wave;", w, acc[7] / 256);
    for (int i = 0; i < 7; ++i) printf("  %s %.0f", names[i], acc[i] / acc[7]);
    printf("  (cycles per full step; the first three also count the peeled steps)\n");
    const char* bn[6] = {"prologue", "step 0", "full steps", "last step", "idle steps + barrier", "finish"};
    printf("        %.0f query blocks per wave; cycles per query block:", acc[14] / 256);
    for (int i = 0; i < 6; ++i) printf("  %s %.0f", bn[i], acc[8 + i] / acc[14]);
    printf("\n        whole kernel: %.0f cycles in %.1f us (%.3f GHz)\n", acc[15] / 256, acc[16] / 256 / 100.0, acc[15] / acc[16] / 10.0);
  }
  // spread of the workgroups' lifetimes (wave 0 of each)
  std::vector<double> life;
  for (int blk = 0; blk < 256; ++blk) life.push_back(st[blk * 4][16] / 100.0);
  std::sort(life.begin(), life.end());
  printf("workgroup lifetime (first 256 workgroups): min %.1f  median %.1f  p90 %.1f  max %.1f us\n", life[0], life[128], life[230], life[255]);
  return 0;
}
